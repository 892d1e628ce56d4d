// fp32-accurate evaluation path ("bf16 x 3"): every GEMM operand is carried as a hi / lo PAIR array --
// hi = bf16(x), lo = bf16(x - hi), interleaved in groups of 32 values (64 bf16: 32 hi, then the 32 lo of the same
// values) so that one 128-byte line holds both halves of a K step -- and a product is evaluated as
//     x w  ~=  hi_x hi_w + hi_x lo_w + lo_x hi_w        (three MFMAs per fragment pair, one f32 accumulator)
// inside ONE GEMM launch (vp_gemm_args.split3), which keeps ~16 significand bits per operand (relative error ~1e-5
// per product instead of bf16's 4e-3).  Everything between GEMMs (bias, GELU, residual, BatchNorm scale/shift,
// ReLU, softmax, LayerNorm) is f32 and fused into the producing kernel, which writes the next GEMM's pair operand
// directly.  This path exists to show parity with the reference's fp32 CPU forward at the 1e-3 max-abs heat-map
// tolerance on non-degenerate weights.  The kernels here build pair arrays from f32 data.
#include "common.h"
#include "vitpose_hip.h"

namespace {

inline int grid_for(size_t work_items, int cap = 8192) {
    size_t g = (work_items + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g ? g : 1));
}

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = (bf16)a[j];
        hi[4 + j] = (bf16)b[j];
        lo[j] = (bf16)(a[j] - (float)hi[j]);
        lo[4 + j] = (bf16)(b[j] - (float)hi[4 + j]);
    }
}

// pair array of a flat f32 buffer whose tensors start on multiples of 32 elements and have rows of a multiple of
// 32 values (weights): value i goes to (i / 32) * 64 + i % 32 (hi) and 32 further on (lo)
__global__ __launch_bounds__(256) void split_pair_kernel(const float* __restrict__ x, bf16* __restrict__ pair, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8), b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
        bf16x8 hi, lo;
        split8(a, b, hi, lo);
        bf16* o = pair + (i >> 2) * 64 + (i & 3) * 8;
        *reinterpret_cast<bf16x8*>(o) = hi;
        *reinterpret_cast<bf16x8*>(o + 32) = lo;
    }
}

// patch-embedding im2col with the hi / lo split of the f32 pixels, pair layout [tokens][2 * C*p*p]
__global__ __launch_bounds__(256) void patch_im2col_pair_kernel(const float* __restrict__ img, bf16* __restrict__ pair,
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
        bf16x8 oh, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int xx = px * stride - pad + kx0 + j;
            const bool ok = y >= 0 && y < H && xx >= 0 && xx < W;
            const float v = ok ? img[(((size_t)b * C + c) * H + y) * W + xx] : 0.f;
            oh[j] = (bf16)v;
            ol[j] = (bf16)(v - (float)oh[j]);
        }
        bf16* o = pair + tok * 2 * K + ((k >> 5) << 6) + (k & 31);
        *reinterpret_cast<bf16x8*>(o) = oh;
        *reinterpret_cast<bf16x8*>(o + 32) = ol;
    }
}

}  // namespace

extern "C" int vp_split_pair_f32(const float* x, void* pair, size_t n, hipStream_t stream) {
    VP_CHECK_ARG(x && pair && n > 0 && n % 32 == 0);
    VP_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)pair & 15) == 0);
    hipLaunchKernelGGL(split_pair_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, x, (bf16*)pair, n / 8);
    VP_LAUNCH_RET();
}

extern "C" int vp_patch_im2col_pair(const float* img, void* col_pair, int B, int C, int H, int W, int patch, int stride,
                                    int pad, hipStream_t stream) {
    VP_CHECK_ARG(img && col_pair && B > 0 && C > 0 && patch % 8 == 0 && stride > 0 && (C * patch * patch) % 32 == 0);
    const int Hp = (H + 2 * pad - patch) / stride + 1, Wp = (W + 2 * pad - patch) / stride + 1;
    VP_CHECK_ARG(Hp > 0 && Wp > 0);
    const size_t total = (size_t)B * Hp * Wp * (C * patch * patch / 8);
    hipLaunchKernelGGL(patch_im2col_pair_kernel, dim3(grid_for(total)), dim3(256), 0, stream, img, (bf16*)col_pair, B, C,
                       H, W, Hp, Wp, patch, stride, pad);
    VP_LAUNCH_RET();
}
