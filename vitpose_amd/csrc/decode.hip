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
