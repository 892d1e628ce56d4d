// Heat map -> keypoint decoding on the device (the reference does this on the host with numpy + cv2 after
// copying every heat map over PCIe): one workgroup per (image, joint) map held in LDS; only
// [N*K] x (x, y, score) leave the GPU.
#include "common.h"
#include "vitpose_hip.h"

namespace {

struct BlurTaps {
    float k[VP_DECODE_MAX_KERNEL];
    int n;
};

__device__ __forceinline__ int reflect101(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// mode 0: arg-max only; 1: quarter-pixel shift towards the higher neighbour; 2: DARK-UDP Newton step on
// the blurred, clipped, log heat map.
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ heat, float* __restrict__ preds,
                                                     float* __restrict__ maxvals, int H, int W, int mode,
                                                     BlurTaps bk) {
    extern __shared__ float hm[];                 // [H*W] map, then scratch
    const int HW = H * W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* rowdot = hm + HW;                      // [7][VP_DECODE_MAX_KERNEL]
    float* L = rowdot + 7 * VP_DECODE_MAX_KERNEL; // [7] log values, then [4 waves] x (val, idx)
    float* red = L + 8;
    const float* src = heat + (size_t)blockIdx.x * HW;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int i = tid; i < HW; i += 256) {
        const float v = src[i];
        hm[i] = v;
        if (v > best) { best = v; bidx = i; }     // ascending i: keeps the first maximum of this thread
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane == 0) { red[2 * wave] = best; red[2 * wave + 1] = __int_as_float(bidx); }
    __syncthreads();
    best = red[0];
    bidx = __float_as_int(red[1]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float ov = red[2 * w];
        const int oi = __float_as_int(red[2 * w + 1]);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    // np.argmax semantics (first maximum); a non-positive maximum marks the joint invalid (-1, -1)
    const bool valid = best > 0.f;
    const int px = valid ? bidx % W : -1, py = valid ? bidx / W : -1;
    float outx = (float)px, outy = (float)py;
    if (mode == 1) {
        if (tid == 0 && 1 < px && px < W - 1 && 1 < py && py < H - 1) {
            const float ddx = hm[py * W + px + 1] - hm[py * W + px - 1];
            const float ddy = hm[(py + 1) * W + px] - hm[(py - 1) * W + px];
            outx += ddx > 0.f ? 0.25f : (ddx < 0.f ? -0.25f : 0.f);
            outy += ddy > 0.f ? 0.25f : (ddy < 0.f ? -0.25f : 0.f);
        }
    } else if (mode == 2) {
        // stencil taps of the edge-padded log map: (0,0) (1,0) (0,1) (1,1) (-1,-1) (-1,0) (0,-1)
        const int ox[7] = {0, 1, 0, 1, -1, -1, 0}, oy[7] = {0, 0, 1, 1, -1, 0, -1};
        const int n = bk.n, R = n >> 1;
        if (tid < 7 * n) {
            const int pos = tid / n, i = tid - pos * n;
            const int x = min(max(px + ox[pos], 0), W - 1), y = min(max(py + oy[pos], 0), H - 1);
            const float* row = hm + reflect101(y + i - R, H) * W;
            float acc = 0.f;
            for (int j = 0; j < n; ++j) acc += bk.k[j] * row[reflect101(x + j - R, W)];
            rowdot[pos * VP_DECODE_MAX_KERNEL + i] = acc;
        }
        __syncthreads();
        if (tid < 7) {
            float acc = 0.f;
            for (int i = 0; i < n; ++i) acc += bk.k[i] * rowdot[tid * VP_DECODE_MAX_KERNEL + i];
            L[tid] = logf(fminf(fmaxf(acc, 0.001f), 50.f));
        }
        __syncthreads();
        if (tid == 0) {
            const float i_ = L[0], ix1 = L[1], iy1 = L[2], ix1y1 = L[3], ix1_y1_ = L[4], ix1_ = L[5], iy1_ = L[6];
            const float dx = 0.5f * (ix1 - ix1_), dy = 0.5f * (iy1 - iy1_);
            const float dxx = ix1 - 2.f * i_ + ix1_, dyy = iy1 - 2.f * i_ + iy1_;
            const float dxy = 0.5f * (ix1y1 - ix1 - iy1 + i_ + i_ - ix1_ - iy1_ + ix1_y1_);
            const double eps = 1.1920928955078125e-07;   // np.finfo(np.float32).eps
            const double a = (double)dxx + eps, d = (double)dyy + eps, b = (double)dxy;
            const double det = a * d - b * b;
            outx = (float)((double)outx - (d * (double)dx - b * (double)dy) / det);
            outy = (float)((double)outy - (a * (double)dy - b * (double)dx) / det);
        }
    }
    if (tid == 0) {
        preds[2 * (size_t)blockIdx.x] = outx;
        preds[2 * (size_t)blockIdx.x + 1] = outy;
        maxvals[blockIdx.x] = best;
    }
}

// first maximum of v over the workgroup (np.argmax order); every thread returns the result
__device__ __forceinline__ void block_argmax(float& best, int& bidx, float* red, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    __syncthreads();                                // `red` may still be read from an earlier reduction
    if (lane == 0) { red[2 * wave] = best; red[2 * wave + 1] = __int_as_float(bidx); }
    __syncthreads();
    best = red[0];
    bidx = __float_as_int(red[1]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float ov = red[2 * w];
        const int oi = __float_as_int(red[2 * w + 1]);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
}

// The branches of keypoints_from_heatmaps that blur whole maps, one workgroup per joint:
//   mode 3 'unbiased' (DARK): arg-max of the map; _gaussian_blur (blur inside a zero border, rescaled so that the
//          maximum is restored), log(max(., 1e-10)), second-order Taylor step (_taylor) at the arg-max;
//   mode 4 'megvii': _gaussian_blur first, then arg-max, +-0.25 shift and +0.5 on the modulated map;
//          score = max / 255 + 0.5;
//   mode 5 UDP 'CombinedTarget': a joint owns three maps (heat, x offset, y offset): the heat map is blurred with
//          2 * kernel + 1 taps (taps2), the offset maps with `kernel` taps (BORDER_REFLECT_101), arg-max of the
//          blurred heat map, + valid_radius * (blurred offsets at the arg-max).
__global__ __launch_bounds__(256) void decode_blur_kernel(const float* __restrict__ heat, float* __restrict__ preds,
                                                          float* __restrict__ maxvals, int H, int W, int mode,
                                                          BlurTaps bk, BlurTaps bk2, float valid_radius) {
    extern __shared__ float sm[];
    const int HW = H * W, tid = threadIdx.x;
    float* a = sm;                  // the map
    float* t = a + HW;              // row pass
    float* b = t + HW;              // blurred map
    float* red = b + HW;            // [8] reduction scratch, then [2][VP_DECODE_MAX_KERNEL] row dots, [2] offsets
    float* rowdot = red + 8;
    float* offs = rowdot + 2 * VP_DECODE_MAX_KERNEL;
    const int maps_per_joint = mode == 5 ? 3 : 1;
    const float* src = heat + (size_t)blockIdx.x * maps_per_joint * HW;
    float best0 = -INFINITY;
    int idx0 = 0x7fffffff;
    for (int i = tid; i < HW; i += 256) {
        const float v = src[i];
        a[i] = v;
        if (v > best0) { best0 = v; idx0 = i; }
    }
    block_argmax(best0, idx0, red, tid);            // (syncs: `a` is complete)
    const BlurTaps& k0 = mode == 5 ? bk2 : bk;
    const int n = k0.n, R = n >> 1;
    const bool zero_border = mode != 5;
    for (int i = tid; i < HW; i += 256) {
        const int y = i / W, x = i - y * W;
        float acc = 0.f;
        for (int j = 0; j < n; ++j) {
            const int xx = x + j - R;
            const float v = zero_border ? ((xx >= 0 && xx < W) ? a[y * W + xx] : 0.f) : a[y * W + reflect101(xx, W)];
            acc += k0.k[j] * v;
        }
        t[i] = acc;
    }
    __syncthreads();
    float bbest = -INFINITY;
    int bidx = 0x7fffffff;
    for (int i = tid; i < HW; i += 256) {
        const int y = i / W, x = i - y * W;
        float acc = 0.f;
        for (int j = 0; j < n; ++j) {
            const int yy = y + j - R;
            const float v = zero_border ? ((yy >= 0 && yy < H) ? t[yy * W + x] : 0.f) : t[reflect101(yy, H) * W + x];
            acc += k0.k[j] * v;
        }
        b[i] = acc;
        if (acc > bbest) { bbest = acc; bidx = i; }
    }
    block_argmax(bbest, bidx, red, tid);            // (syncs: `b` is complete)
    float outx, outy, score;
    if (mode == 5) {
        const bool valid = bbest > 0.f;
        // A joint without a positive maximum has the arg-max (-1, -1) (_get_max_preds), and the reference still looks
        // its offsets up, at flat index -1 - W + W * H * j of the flattened offset maps (top_down_eval.py:582-585):
        // that is pixel (H - 2, W - 1) of the PREVIOUS joint's offset maps, and for the very first joint numpy's
        // negative index wraps to the last joint of the batch.  Same lookup here.
        const int px = valid ? bidx % W : W - 1, py = valid ? bidx / W : H - 2;
        const int owner = valid ? (int)blockIdx.x : (blockIdx.x == 0 ? (int)gridDim.x - 1 : (int)blockIdx.x - 1);
        const float* osrc = heat + (size_t)owner * 3 * HW;
        const int m = bk.n, r = m >> 1;
        if (tid < 2 * m) {
            const int which = tid / m, i = tid - which * m;
            const float* row = osrc + (size_t)(1 + which) * HW + reflect101(py + i - r, H) * W;
            float acc = 0.f;
            for (int j = 0; j < m; ++j) acc += bk.k[j] * row[reflect101(px + j - r, W)];
            rowdot[which * VP_DECODE_MAX_KERNEL + i] = acc;
        }
        __syncthreads();
        if (tid < 2) {
            float acc = 0.f;
            for (int i = 0; i < m; ++i) acc += bk.k[i] * rowdot[tid * VP_DECODE_MAX_KERNEL + i];
            offs[tid] = acc * valid_radius;
        }
        __syncthreads();
        outx = (valid ? (float)px : -1.f) + offs[0];
        outy = (valid ? (float)py : -1.f) + offs[1];
        score = bbest;
    } else {
        // _gaussian_blur: the blurred map is rescaled so that its maximum equals the original one
        const float scale = best0 / bbest;
        if (mode == 4) {
            float mbest = -INFINITY;
            int midx = 0x7fffffff;
            for (int i = tid; i < HW; i += 256) {
                const float v = b[i] * scale;
                b[i] = v;
                if (v > mbest) { mbest = v; midx = i; }
            }
            block_argmax(mbest, midx, red, tid);
            const bool valid = mbest > 0.f;
            const int px = valid ? midx % W : -1, py = valid ? midx / W : -1;
            outx = (float)px;
            outy = (float)py;
            if (1 < px && px < W - 1 && 1 < py && py < H - 1) {
                const float ddx = b[py * W + px + 1] - b[py * W + px - 1];
                const float ddy = b[(py + 1) * W + px] - b[(py - 1) * W + px];
                outx += (ddx > 0.f ? 0.25f : (ddx < 0.f ? -0.25f : 0.f)) + 0.5f;
                outy += (ddy > 0.f ? 0.25f : (ddy < 0.f ? -0.25f : 0.f)) + 0.5f;
            }
            score = mbest / 255.0f + 0.5f;
        } else {
            const bool valid = best0 > 0.f;
            const int px = valid ? idx0 % W : -1, py = valid ? idx0 / W : -1;
            outx = (float)px;
            outy = (float)py;
            score = best0;
            if (1 < px && px < W - 2 && 1 < py && py < H - 2) {
                auto L = [&](int yy, int xx) { return logf(fmaxf(b[yy * W + xx] * scale, 1e-10f)); };
                const float c0 = L(py, px);
                const float dx = 0.5f * (L(py, px + 1) - L(py, px - 1)), dy = 0.5f * (L(py + 1, px) - L(py - 1, px));
                const float dxx = 0.25f * (L(py, px + 2) - 2.f * c0 + L(py, px - 2));
                const float dxy = 0.25f * (L(py + 1, px + 1) - L(py - 1, px + 1) - L(py + 1, px - 1) + L(py - 1, px - 1));
                const float dyy = 0.25f * (L(py + 2, px) - 2.f * c0 + L(py - 2, px));
                const float det = dxx * dyy - dxy * dxy;
                if (det != 0.f) {
                    outx -= (dyy * dx - dxy * dy) / det;
                    outy -= (dxx * dy - dxy * dx) / det;
                }
            }
        }
    }
    if (tid == 0) {
        preds[2 * (size_t)blockIdx.x] = outx;
        preds[2 * (size_t)blockIdx.x + 1] = outy;
        maxvals[blockIdx.x] = score;
    }
}

__global__ __launch_bounds__(256) void flip_back_avg_kernel(const float* __restrict__ heat,
                                                            const float* __restrict__ flipped,
                                                            const int* __restrict__ perm, int shift,
                                                            float* __restrict__ out, int K, int H, int W, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const size_t r = i / W;                       // (n * K + k) * H + y
        const int y = (int)(r % H);
        const size_t nk = r / H;
        const int k = (int)(nk % K);
        const size_t n = nk / K;
        const int xs = shift ? max(x - 1, 0) : x;
        const float fb = flipped[((n * K + perm[k]) * H + y) * W + (W - 1 - xs)];
        out[i] = (heat[i] + fb) * 0.5f;
    }
}

}  // namespace

extern "C" int vp_decode_heatmaps(const float* heat, float* preds, float* maxvals, int maps, int H, int W, int mode,
                                  const float* taps_host, int ntaps, hipStream_t stream) {
    VP_CHECK_ARG(heat && preds && maxvals && maps > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 2);
    BlurTaps bk{};
    if (mode == 2) {
        VP_CHECK_ARG(taps_host && ntaps > 0 && ntaps % 2 == 1 && ntaps <= VP_DECODE_MAX_KERNEL);
        VP_CHECK_ARG(H > ntaps / 2 && W > ntaps / 2);      // one reflection at the border
        for (int i = 0; i < ntaps; ++i) bk.k[i] = taps_host[i];
        bk.n = ntaps;
    }
    const size_t lds = ((size_t)H * W + 7 * VP_DECODE_MAX_KERNEL + 16) * sizeof(float);
    if (lds > 150 * 1024) return VP_ERR_UNSUPPORTED;
    (void)hipFuncSetAttribute((const void*)decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(decode_kernel, dim3(maps), dim3(256), lds, stream, heat, preds, maxvals, H, W, mode, bk);
    VP_LAUNCH_RET();
}

extern "C" int vp_flip_back_avg(const float* heat, const float* heat_flipped, const int* perm, int shift, float* out,
                                int N, int K, int H, int W, hipStream_t stream) {
    VP_CHECK_ARG(heat && heat_flipped && perm && out && N > 0 && K > 0 && H > 0 && W > 0);
    const size_t total = (size_t)N * K * H * W;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(flip_back_avg_kernel, dim3(grid), dim3(256), 0, stream, heat, heat_flipped, perm, shift, out, K,
                       H, W, total);
    VP_LAUNCH_RET();
}

extern "C" int vp_decode_blurred(const float* heat, float* preds, float* maxvals, int joints, int H, int W, int mode,
                                 const float* taps_host, int ntaps, const float* taps2_host, int ntaps2,
                                 float valid_radius, hipStream_t stream) {
    VP_CHECK_ARG(heat && preds && maxvals && joints > 0 && H > 0 && W > 0 && mode >= 3 && mode <= 5);
    VP_CHECK_ARG(taps_host && ntaps > 0 && ntaps % 2 == 1 && ntaps <= VP_DECODE_MAX_KERNEL);
    BlurTaps bk{}, bk2{};
    for (int i = 0; i < ntaps; ++i) bk.k[i] = taps_host[i];
    bk.n = ntaps;
    if (mode == 5) {
        VP_CHECK_ARG(taps2_host && ntaps2 > 0 && ntaps2 % 2 == 1 && ntaps2 <= VP_DECODE_MAX_KERNEL);
        VP_CHECK_ARG(H > ntaps2 / 2 && W > ntaps2 / 2 && H > ntaps / 2 && W > ntaps / 2);   // one reflection at the border
        for (int i = 0; i < ntaps2; ++i) bk2.k[i] = taps2_host[i];
        bk2.n = ntaps2;
    }
    const size_t lds = (3 * (size_t)H * W + 8 + 2 * VP_DECODE_MAX_KERNEL + 8) * sizeof(float);
    if (lds > 150 * 1024) return VP_ERR_UNSUPPORTED;
    (void)hipFuncSetAttribute((const void*)decode_blur_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(decode_blur_kernel, dim3(joints), dim3(256), lds, stream, heat, preds, maxvals, H, W, mode, bk, bk2,
                       valid_radius);
    VP_LAUNCH_RET();
}
