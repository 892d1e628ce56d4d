// HBM-bound data-movement kernels around the GEMMs of the ViTPose hot path:
// im2col views (patch embedding, sub-pixel transposed conv), layout/dtype changes,
// column sums.  All are pure streaming kernels (every byte read/written once,
// 8-16 B per lane where the layout allows).
#include "common.h"
#include "vitpose_hip.h"

namespace {

// ---- patch embedding im2col: NCHW f32 image -> [B*Hp*Wp][C*p*p] bf16 rows (k = (c, ky, kx)) ----
// One thread per 8 consecutive kx of one (token, c, ky): 8 pixels of one image row.
__global__ __launch_bounds__(256) void patch_im2col_kernel(const float* __restrict__ img, bf16* __restrict__ col,
                                                           int B, int C, int H, int W, int Hp, int Wp, int p,
                                                           int stride, int pad) {
    const int K = C * p * p;
    const int chunks = K >> 3;
    const size_t total = (size_t)B * Hp * Wp * chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const size_t tok = i / chunks;
        const int k = ch << 3;
        const int c = k / (p * p), ky = (k / p) % p, kx0 = k % p;
        const int px = (int)(tok % Wp), py = (int)((tok / Wp) % Hp), b = (int)(tok / ((size_t)Wp * Hp));
        const int y = py * stride - pad + ky;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = px * stride - pad + kx0 + j;
            const bool ok = y >= 0 && y < H && x >= 0 && x < W;
            o[j] = (bf16)(ok ? img[(((size_t)b * C + c) * H + y) * W + x] : 0.f);
        }
        *reinterpret_cast<bf16x8*>(col + tok * K + k) = o;
    }
}

// ---- generic 4-D strided gather + cast: out[i0][i1][i2][i3] (contiguous) = in[sum i_k * s_k] ----
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void strided_cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, int d0,
                                                           int d1, int d2, int d3, long s0, long s1, long s2, long s3) {
    const size_t total = (size_t)d0 * d1 * d2 * d3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int i3 = (int)(i % d3);
        size_t r = i / d3;
        const int i2 = (int)(r % d2);
        r /= d2;
        const int i1 = (int)(r % d1);
        const int i0 = (int)(r / d1);
        out[i] = (TO)(float)in[i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
    }
}

// ---- compute layouts of a ConvTranspose2d(k=4) weight W [Ci][Co][4][4] f32, in one pass ----
//   wph[py*2+px][co][(ty*2+tx)*Ci + ci] = W[ci][co][(1-py)+2ty][(1-px)+2tx]   (sub-pixel phase matrices, forward)
//   wd[ci][(ky*4+kx)*Co + co]          = W[ci][co][ky][kx]                     (16-tap matrix, data gradient)
// One workgroup per 32 x 32 (ci, co) patch: read with co fastest (64 contiguous bytes per thread), stage as bf16
// in LDS, write wph with ci fastest and wd with co fastest.
__global__ __launch_bounds__(256) void deconv_weight_prep_kernel(const float* __restrict__ w, bf16* __restrict__ wph,
                                                                 bf16* __restrict__ wd, int Ci, int Co) {
    __shared__ bf16 tile[32][33][16];
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int cl = i >> 5, ol = i & 31;           // co fastest
        const int ci = ci0 + cl, co = co0 + ol;
        if (ci < Ci && co < Co) {
            const f32x4* src = reinterpret_cast<const f32x4*>(w + ((size_t)ci * Co + co) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = src[q];
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[cl][ol][q * 4 + r] = (bf16)v[r];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) {
        {   // wd: co fastest
            const int cl = i >> 5, ol = i & 31;
            const int ci = ci0 + cl, co = co0 + ol;
            if (ci < Ci && co < Co) {
#pragma unroll
                for (int t = 0; t < 16; ++t) wd[(size_t)ci * 16 * Co + (size_t)t * Co + co] = tile[cl][ol][t];
            }
        }
        {   // wph: ci fastest
            const int ol = i >> 5, cl = i & 31;
            const int ci = ci0 + cl, co = co0 + ol;
            if (ci < Ci && co < Co) {
#pragma unroll
                for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int ky = (1 - (ph >> 1)) + 2 * (t >> 1), kx = (1 - (ph & 1)) + 2 * (t & 1);
                        wph[((size_t)ph * Co + co) * 4 * Ci + (size_t)t * Ci + ci] = tile[cl][ol][ky * 4 + kx];
                    }
            }
        }
    }
}

// ---- column sums of a bf16 [M][N] matrix into f32 [N] (atomic accumulate) ----
// block = 256 threads = 32 column-chunks (8 cols) x 8 row lanes; grid.x tiles columns, grid.y rows.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ a, float* __restrict__ out, int M, int N,
                                                     int lda, int rows_per_block) {
    __shared__ float red[8][256 + 8];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int col = (blockIdx.x * 32 + cx) * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < N) {
        const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
        for (int r = r0 + ry; r < r1; r += 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(a + (size_t)r * lda + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[ry][cx * 8 + j] = s[j];
    __syncthreads();
    const int c = threadIdx.x;  // 256 columns of this block
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][c];
    const int gc = blockIdx.x * 256 + c;
    if (gc < N) atomicAdd(out + gc, t);
}

// ---- sum over the batch of token-major f32 [B][T*E] -> [T*E] (pos_embed gradient) ----
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B,
                                                        size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= B; b += 8) {           // eight images in flight (few workgroups: latency-, not bandwidth-bound)
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(x + (size_t)(b + j) * n + i);
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; b < B; ++b) s += *reinterpret_cast<const f32x4*>(x + (size_t)b * n + i);
    *reinterpret_cast<f32x4*>(out + i) = s;
}

// ---- sub-pixel view of a k4/s2/p1 transposed conv, output parity (py, px) ----
// col[(b,y,x)][t*C + c] = in[b][y + dy(py,ty)][x + dx(px,tx)][c]  (0 outside), t = ty*2 + tx,
// dy(0,{0,1}) = {0,-1}, dy(1,{0,1}) = {+1,0}; the matching kernel taps are ky(0,.) = {1,3}, ky(1,.) = {0,2}.
__global__ __launch_bounds__(256) void deconv_phase_im2col_kernel(const bf16* __restrict__ in, bf16* __restrict__ col,
                                                                  int B, int H, int W, int C, int py, int px) {
    const int chunks = (4 * C) >> 3;
    const size_t total = (size_t)B * H * W * chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const size_t pix = i / chunks;
        const int k = ch << 3, t = k / C, c = k % C;
        const int ty = t >> 1, tx = t & 1;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
        const int yy = y + (py ? 1 - ty : -ty), xx = x + (px ? 1 - tx : -tx);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            v = *reinterpret_cast<const uint4*>(in + (((size_t)b * H + yy) * W + xx) * C + c);
        *reinterpret_cast<uint4*>(col + pix * (size_t)(4 * C) + k) = v;
    }
}

// ---- 16-tap gather of the transposed conv's OUTPUT grid around each INPUT pixel ----
// G[(b,iy,ix)][(ky*4+kx)*C + c] = dout[b][2iy-1+ky][2ix-1+kx][c]  (0 outside); H, W = input grid.
// Both the data gradient (G x Wd^T) and the weight gradient (in^T x G) read it.
__global__ __launch_bounds__(256) void deconv_out_gather16_kernel(const bf16* __restrict__ dout, bf16* __restrict__ g,
                                                                  int B, int H, int W, int C) {
    // one wave per input pixel: the pixel's coordinates are decoded once (wave-uniform), the lanes walk its
    // 16 * C / 8 sixteen-byte chunks with 32-bit index arithmetic
    const unsigned chunks = (16u * (unsigned)C) >> 3, cpt = (unsigned)C >> 3;   // per pixel, per tap
    const size_t npix = (size_t)B * H * W;
    const int Ho = 2 * H, Wo = 2 * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t pix = (size_t)blockIdx.x * 4 + wave; pix < npix; pix += (size_t)gridDim.x * 4) {
        const unsigned p32 = (unsigned)(pix % ((size_t)W * H));
        const size_t b = pix / ((size_t)W * H);
        const int y = (int)(p32 / (unsigned)W), x = (int)(p32 % (unsigned)W);
        const bf16* src = dout + b * (size_t)Ho * Wo * C;
        bf16* dst = g + pix * (size_t)(16 * C);
        for (unsigned ch = lane; ch < chunks; ch += 64) {
            const unsigned t = ch / cpt, c = (ch - t * cpt) << 3;
            const int oy = 2 * y - 1 + (int)(t >> 2), ox = 2 * x - 1 + (int)(t & 3);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo)
                v = *reinterpret_cast<const uint4*>(src + ((size_t)oy * Wo + ox) * C + c);
            *reinterpret_cast<uint4*>(dst + ((size_t)ch << 3)) = v;
        }
    }
}

// ---- out[i] (+)= sum over the split-K slabs ----
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int nsplit, long stride,
                                                            size_t n4, float* __restrict__ out, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 s = accumulate ? *reinterpret_cast<const f32x4*>(out + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        // four slabs in flight per step (a load that the next add waits for costs a full memory latency)
        int z = 0;
        for (; z + 4 <= nsplit; z += 4) {
            const float* q = slab + (size_t)z * stride + i * 4;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(q), a1 = *reinterpret_cast<const f32x4*>(q + stride);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(q + 2 * stride);
            const f32x4 a3 = *reinterpret_cast<const f32x4*>(q + 3 * stride);
            s += (a0 + a1) + (a2 + a3);
        }
        for (; z < nsplit; ++z) s += *reinterpret_cast<const f32x4*>(slab + (size_t)z * stride + i * 4);
        *reinterpret_cast<f32x4*>(out + i * 4) = s;
    }
}

inline int grid_for(size_t work_items) {
    size_t g = (work_items + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g ? g : 1));
}

}  // namespace

// Non-overlapping 16 x 16 patches (stride == patch: every ViTPose configuration): one workgroup per row of patches.
// The 16 image rows of that strip are read as whole rows (coalesced; two pixels per lane), converted, scattered into
// the strip's output image in LDS -- Wp consecutive tokens x K columns are one contiguous piece of `col` -- and
// written out 16 bytes per lane.  The general kernel above reads 64-byte pieces; this one moves the same bytes ~3x
// faster.
constexpr int PATCH_STRIP_MAX = 12 * 3 * 256;     // bf16 elements of one strip in LDS (Wp * C * 256)
template <int C, int NSUB>     // channels; strip rows in flight per pass (256 / (8 Wp))
__global__ __launch_bounds__(256) void patch_im2col_strip_kernel(const float* __restrict__ img, bf16* __restrict__ col,
                                                                 int H, int W, int Hp, int Wp, int pad) {
    __shared__ __attribute__((aligned(16))) bf16 tile[PATCH_STRIP_MAX];
    const int b = blockIdx.x / Hp, py = blockIdx.x % Hp;
    const int K = C * 256;
    const int pairs = Wp * 8;                      // pixel pairs per strip row
    const int rsub = threadIdx.x / pairs, xp = threadIdx.x % pairs;
    if (rsub < NSUB) {
        const int x = xp * 2 - pad;                // first pixel of the pair (pad is even: the pair never straddles x = 0)
        const int px = xp >> 3, kx = (xp & 7) * 2;
        const bool xok = x >= 0 && x + 1 < W;
#pragma unroll 8
        for (int it = 0; it < (C * 16 + NSUB - 1) / NSUB; ++it) {
            const int r = rsub + it * NSUB;
            if (r >= C * 16) break;
            const int c = r >> 4, ky = r & 15;
            const int y = py * 16 - pad + ky;
            float v0 = 0.f, v1 = 0.f;
            if (y >= 0 && y < H) {
                const float* row = img + (((size_t)b * C + c) * H + y) * W;
                if (xok) {
                    const float2 v = *reinterpret_cast<const float2*>(row + x);
                    v0 = v.x;
                    v1 = v.y;
                } else {
                    if (x >= 0 && x < W) v0 = row[x];
                    if (x + 1 >= 0 && x + 1 < W) v1 = row[x + 1];
                }
            }
            bf16* d = tile + (px * C + c) * 256 + ky * 16 + kx;
            d[0] = (bf16)v0;
            d[1] = (bf16)v1;
        }
    }
    __syncthreads();
    const int chunks = Wp * K / 8;
    uint4* dst = reinterpret_cast<uint4*>(col + (size_t)blockIdx.x * Wp * K);
    for (int i = threadIdx.x; i < chunks; i += 256) dst[i] = *reinterpret_cast<const uint4*>(tile + i * 8);
}

extern "C" int vp_patch_im2col(const float* img, void* col_bf16, int B, int C, int H, int W, int patch, int stride,
                               int pad, hipStream_t stream) {
    VP_CHECK_ARG(img && col_bf16 && B > 0 && C > 0 && patch % 8 == 0 && stride > 0);
    const int Hp = (H + 2 * pad - patch) / stride + 1, Wp = (W + 2 * pad - patch) / stride + 1;
    VP_CHECK_ARG(Hp > 0 && Wp > 0);
    if (patch == 16 && stride == 16 && pad >= 0 && pad % 2 == 0 && W % 2 == 0 && C == 3 && 256 / (Wp * 8) == 2 &&
        Wp * C * 256 <= PATCH_STRIP_MAX && ((uintptr_t)img & 7) == 0 && ((uintptr_t)col_bf16 & 15) == 0) {
        hipLaunchKernelGGL((patch_im2col_strip_kernel<3, 2>), dim3(B * Hp), dim3(256), 0, stream, img, (bf16*)col_bf16, H,
                           W, Hp, Wp, pad);
        VP_LAUNCH_RET();
    }
    const size_t total = (size_t)B * Hp * Wp * (C * patch * patch / 8);
    hipLaunchKernelGGL(patch_im2col_kernel, dim3(grid_for(total)), dim3(256), 0, stream, img, (bf16*)col_bf16, B, C, H,
                       W, Hp, Wp, patch, stride, pad);
    VP_LAUNCH_RET();
}

extern "C" int vp_strided_cast(const void* in, int in_dtype, void* out, int out_dtype, const int dims[4],
                               const long strides[4], hipStream_t stream) {
    VP_CHECK_ARG(in && out && dims && strides);
    const size_t total = (size_t)dims[0] * dims[1] * dims[2] * dims[3];
    VP_CHECK_ARG(total > 0);
    dim3 grid(grid_for(total)), block(256);
#define VP_SC(TI, TO)                                                                                     \
    hipLaunchKernelGGL((strided_cast_kernel<TI, TO>), grid, block, 0, stream, (const TI*)in, (TO*)out,    \
                       dims[0], dims[1], dims[2], dims[3], strides[0], strides[1], strides[2], strides[3])
    if (in_dtype == VP_F32 && out_dtype == VP_BF16) VP_SC(float, bf16);
    else if (in_dtype == VP_F32 && out_dtype == VP_F32) VP_SC(float, float);
    else if (in_dtype == VP_BF16 && out_dtype == VP_F32) VP_SC(bf16, float);
    else if (in_dtype == VP_BF16 && out_dtype == VP_BF16) VP_SC(bf16, bf16);
    else return VP_ERR_ARG;
#undef VP_SC
    VP_LAUNCH_RET();
}

// Narrow matrices (N <= 64 columns, rows packed: lda == N; the padded heat-map gradient): the kernel above would keep
// one thread in eight busy.  Every thread walks 16-byte chunks of the flat array with a stride that is a multiple
// of the chunks per row, so it stays on one column group; a block sums its threads per column group.
template <int CG>     // 16-byte chunks per row
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const bf16* __restrict__ a, float* __restrict__ out, size_t n8) {
    __shared__ float red[256][9];
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;       // 256 % CG == 0
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(a + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = s[j];
    __syncthreads();
    if (threadIdx.x < CG * 8) {
        const int cg = threadIdx.x >> 3, j = threadIdx.x & 7;
        float t = 0.f;
        for (int r = cg; r < 256; r += CG) t += red[r][j];
        atomicAdd(out + cg * 8 + j, t);
    }
}

extern "C" int vp_colsum_bf16(const void* a, float* out, int M, int N, int lda, hipStream_t stream) {
    VP_CHECK_ARG(a && out && M > 0 && N > 0 && N % 8 == 0 && lda % 8 == 0);
    if (lda == N && (N == 8 || N == 16 || N == 32 || N == 64) && ((uintptr_t)a & 15) == 0) {
        const size_t n8 = (size_t)M * (N / 8);
        size_t blocks = (n8 + 2047) / 2048;          // >= 8 chunks per thread; few blocks: each ends in N atomics on
        if (blocks > 256) blocks = 256;              // the same N addresses, which serialise
        if (blocks < 1) blocks = 1;
        const dim3 g((unsigned)blocks), b(256);
        switch (N / 8) {
            case 1: hipLaunchKernelGGL(colsum_narrow_kernel<1>, g, b, 0, stream, (const bf16*)a, out, n8); break;
            case 2: hipLaunchKernelGGL(colsum_narrow_kernel<2>, g, b, 0, stream, (const bf16*)a, out, n8); break;
            case 4: hipLaunchKernelGGL(colsum_narrow_kernel<4>, g, b, 0, stream, (const bf16*)a, out, n8); break;
            default: hipLaunchKernelGGL(colsum_narrow_kernel<8>, g, b, 0, stream, (const bf16*)a, out, n8); break;
        }
        VP_LAUNCH_RET();
    }
    const int gx = (N + 255) / 256;
    int gy = (M + 255) / 256;
    if (gy > 512) gy = 512;
    const int rpb = (M + gy - 1) / gy;
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, stream, (const bf16*)a, out, M, N, lda, rpb);
    VP_LAUNCH_RET();
}

extern "C" int vp_splitk_reduce(const float* slab, int nsplit, long stride, size_t n, float* out, int accumulate,
                                hipStream_t stream) {
    VP_CHECK_ARG(slab && out && nsplit > 0 && n > 0 && n % 4 == 0 && stride % 4 == 0 && (size_t)stride >= n);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, slab, nsplit, stride, n / 4,
                       out, accumulate);
    VP_LAUNCH_RET();
}

extern "C" int vp_batch_sum_f32(const float* x, float* out, int B, size_t n, hipStream_t stream) {
    VP_CHECK_ARG(x && out && B > 0 && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, x, out, B, n);
    VP_LAUNCH_RET();
}

extern "C" int vp_deconv_phase_im2col(const void* in, void* col, int B, int H, int W, int C, int py, int px,
                                      hipStream_t stream) {
    VP_CHECK_ARG(in && col && B > 0 && H > 0 && W > 0 && C % 8 == 0 && (py | 1) == 1 && (px | 1) == 1);
    const size_t total = (size_t)B * H * W * (4 * C / 8);
    hipLaunchKernelGGL(deconv_phase_im2col_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16*)in,
                       (bf16*)col, B, H, W, C, py, px);
    VP_LAUNCH_RET();
}

extern "C" int vp_deconv_out_gather16(const void* dout, void* g, int B, int H, int W, int C, hipStream_t stream) {
    VP_CHECK_ARG(dout && g && B > 0 && H > 0 && W > 0 && C % 8 == 0);
    const size_t total = (size_t)B * H * W * (16 * C / 8);
    hipLaunchKernelGGL(deconv_out_gather16_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16*)dout,
                       (bf16*)g, B, H, W, C);
    VP_LAUNCH_RET();
}

// 3 x 3 convolution (stride 1, padding 1) on channels-last maps as a GEMM: col[(b, y, x)][t * C + c] =
// in[b][y + ky - 1][x + kx - 1][c] with t = 3 ky + kx (zero outside the map), and its transpose for the data gradient:
// din[(b, y, x)][c] = sum_t dcol[(b, y - ky + 1, x - kx + 1)][t * C + c] (f32 sum, optional ReLU mask of the layer input).
__global__ __launch_bounds__(256) void im2col3x3_kernel(const bf16* __restrict__ in, bf16* __restrict__ col, int B, int H,
                                                        int W, int C8) {
    const size_t total = (size_t)B * H * W * 9 * C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        const int t = (int)((i / C8) % 9);
        const size_t pix = i / ((size_t)9 * C8);
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        const size_t b = pix / ((size_t)W * H);
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            v = *reinterpret_cast<const uint4*>(in + (((b * H + yy) * W + xx) * (size_t)C8 + c) * 8);
        *reinterpret_cast<uint4*>(col + i * 8) = v;
    }
}

__global__ __launch_bounds__(256) void col2im3x3_kernel(const bf16* __restrict__ dcol, bf16* __restrict__ din,
                                                        const bf16* __restrict__ relu_aux, int B, int H, int W, int C8) {
    const size_t total = (size_t)B * H * W * C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        const size_t pix = i / C8;
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        const size_t b = pix / ((size_t)W * H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y - (t / 3) + 1, xx = x - (t % 3) + 1;      // the output pixel that read this one through tap t
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(dcol + ((((b * H + yy) * W + xx) * 9 + t) * (size_t)C8 + c) * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
            }
        }
        bf16x8 o;
        if (relu_aux) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(relu_aux + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)a[j] > 0.f ? acc[j] : 0.f);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
        }
        *reinterpret_cast<bf16x8*>(din + i * 8) = o;
    }
}

extern "C" int vp_im2col3x3_bf16(const void* in, void* col, int B, int H, int W, int C, hipStream_t stream) {
    VP_CHECK_ARG(in && col && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
    VP_CHECK_ARG((((uintptr_t)in | (uintptr_t)col) & 15) == 0);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for((size_t)B * H * W * 9 * (C / 8))), dim3(256), 0, stream,
                       (const bf16*)in, (bf16*)col, B, H, W, C / 8);
    VP_LAUNCH_RET();
}

extern "C" int vp_col2im3x3_bf16(const void* dcol, void* din, const void* relu_aux, int B, int H, int W, int C,
                                 hipStream_t stream) {
    VP_CHECK_ARG(dcol && din && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
    VP_CHECK_ARG((((uintptr_t)dcol | (uintptr_t)din | (uintptr_t)relu_aux) & 15) == 0);
    hipLaunchKernelGGL(col2im3x3_kernel, dim3(grid_for((size_t)B * H * W * (C / 8))), dim3(256), 0, stream,
                       (const bf16*)dcol, (bf16*)din, (const bf16*)relu_aux, B, H, W, C / 8);
    VP_LAUNCH_RET();
}

// A ConvTranspose2d(stride 2) weight with a 3 x 3 (padding 1, output_padding 1) or 2 x 2 (padding 0) kernel written
// into the 4 x 4 kernel (padding 1) that computes the same layer: output pixel o receives input pixel i through tap
// o - 2 i + 1 of a 4-tap kernel, through tap o - 2 i + 1 of the 3-tap one (its taps are taps 0..2 of the 4) and
// through tap o - 2 i of the 2-tap one (its taps are taps 1..2 of the 4); the other taps are zero.
__global__ __launch_bounds__(256) void deconv_weight_pad4_kernel(const float* __restrict__ w, float* __restrict__ w4,
                                                                 size_t n, int k) {
    const int off = k == 2 ? 1 : 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * 16; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i >> 4;
        const int ky = (int)((i >> 2) & 3) - off, kx = (int)(i & 3) - off;
        w4[i] = (ky >= 0 && ky < k && kx >= 0 && kx < k) ? w[m * k * k + ky * k + kx] : 0.f;
    }
}

extern "C" int vp_deconv_weight_pad4(const float* w, float* w4, size_t n, int ksize, hipStream_t stream) {
    VP_CHECK_ARG(w && w4 && n > 0 && (ksize == 2 || ksize == 3));
    size_t blocks = (n * 16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(deconv_weight_pad4_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, w4, n, ksize);
    VP_LAUNCH_RET();
}

extern "C" int vp_deconv_weight_prep(const float* w, void* wph, void* wd, int Ci, int Co, hipStream_t stream) {
    VP_CHECK_ARG(w && wph && wd && Ci > 0 && Co > 0 && ((uintptr_t)w & 15) == 0);
    hipLaunchKernelGGL(deconv_weight_prep_kernel, dim3((Co + 31) / 32, (Ci + 31) / 32), dim3(256), 0, stream, w,
                       (bf16*)wph, (bf16*)wd, Ci, Co);
    VP_LAUNCH_RET();
}
