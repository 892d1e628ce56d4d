// fp32-accurate evaluation path ("bf16 x 3"): every GEMM operand is carried as a pair of bf16
// arrays hi = bf16(x), lo = bf16(x - hi), and a product is evaluated as
//     x w  ~=  hi_x hi_w + hi_x lo_w + lo_x hi_w        (three MFMA GEMMs, f32 accumulation)
// which keeps ~16 significand bits per operand (relative error ~1e-5 per product instead of
// bf16's 4e-3).  Everything between GEMMs (bias, GELU, residual, BatchNorm scale/shift, ReLU,
// softmax) runs in f32.  This path exists to show parity with the reference's fp32 CPU forward
// at the 1e-3 max-abs heat-map tolerance on non-degenerate weights; it is ~4x the bf16 path's cost.
#include "common.h"
#include "vitpose_hip.h"

namespace {

inline int grid_for(size_t work_items, int cap = 8192) {
    size_t g = (work_items + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g ? g : 1));
}

// y = act(x * colscale + colshift) [* rowscale] [+ resid]; optional outputs: f32 y, bf16 hi / lo.
// One thread per 4 consecutive columns; output rows may be remapped (sub-pixel phases).
__global__ __launch_bounds__(256) void finish_kernel(const float* __restrict__ x, int ldx,
                                                     const float* __restrict__ colscale,
                                                     const float* __restrict__ colshift, int act,
                                                     const float* __restrict__ resid, int ldr,
                                                     const float* __restrict__ rowscale, int rows_per_scale,
                                                     float* __restrict__ out, bf16* __restrict__ hi,
                                                     bf16* __restrict__ lo, int ldo, int M, int N, int row_div,
                                                     int row_hi, int row_lo, int row_off) {
    const int n4 = N >> 2;
    const size_t total = (size_t)M * n4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % n4) * 4, row = (int)(i / n4);
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + c);
        if (colscale) v = v * *reinterpret_cast<const f32x4*>(colscale + c);
        if (colshift) v = v + *reinterpret_cast<const f32x4*>(colshift + c);
        if (act == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752440f));
        } else if (act == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (rowscale) v = v * rowscale[row / rows_per_scale];
        if (resid) v = v + *reinterpret_cast<const f32x4*>(resid + (size_t)row * ldr + c);
        const size_t orow = row_div > 0 ? (size_t)(row / row_div) * row_hi + (row % row_div) * row_lo + row_off
                                        : (size_t)row;
        if (out) *reinterpret_cast<f32x4*>(out + orow * ldo + c) = v;
        if (hi) {
            bf16x4 h = pack4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<bf16x4*>(hi + orow * ldo + c) = h;
            if (lo)
                *reinterpret_cast<bf16x4*>(lo + orow * ldo + c) =
                    pack4(v[0] - (float)h[0], v[1] - (float)h[1], v[2] - (float)h[2], v[3] - (float)h[3]);
        }
    }
}

// hi / lo of a flat f32 buffer (weights)
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, bf16* __restrict__ hi,
                                                    bf16* __restrict__ lo, float* __restrict__ lo_f32, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        const bf16x4 h = pack4(v[0], v[1], v[2], v[3]);
        const f32x4 r = f32x4{v[0] - (float)h[0], v[1] - (float)h[1], v[2] - (float)h[2], v[3] - (float)h[3]};
        if (hi) *reinterpret_cast<bf16x4*>(hi + i * 4) = h;
        if (lo) *reinterpret_cast<bf16x4*>(lo + i * 4) = pack4(r[0], r[1], r[2], r[3]);
        if (lo_f32) *reinterpret_cast<f32x4*>(lo_f32 + i * 4) = r;
    }
}

// f32 attention: one workgroup per (image, head), one thread per query, K and V rows in LDS (every
// thread reads the same row: LDS broadcast), online softmax in f32.
template <int D>
__global__ __launch_bounds__(192) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int H,
                                                       int E, float scale) {
    extern __shared__ float kv[];   // [192][D] keys, then [192][D] values
    float* ks = kv;
    float* vs = kv + 192 * D;
    const int b = blockIdx.x / H, h = blockIdx.x % H, t = threadIdx.x;
    const size_t ld = 3 * (size_t)E;
    const float* base = qkv + (size_t)b * 192 * ld + h * D;
    for (int i = t; i < 192 * (D / 4); i += 192) {
        const int row = i / (D / 4), c = (i % (D / 4)) * 4;
        *reinterpret_cast<f32x4*>(ks + row * D + c) = *reinterpret_cast<const f32x4*>(base + E + row * ld + c);
        *reinterpret_cast<f32x4*>(vs + row * D + c) = *reinterpret_cast<const f32x4*>(base + 2 * E + row * ld + c);
    }
    float q[D], o[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        q[c] = base[(size_t)t * ld + c] * scale;
        o[c] = 0.f;
    }
    __syncthreads();
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j < 192; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) s = fmaf(q[c], ks[j * D + c], s);
        const float mn = fmaxf(m, s);
        const float alpha = expf(m - mn), p = expf(s - mn);
        l = l * alpha + p;
#pragma unroll
        for (int c = 0; c < D; ++c) o[c] = fmaf(p, vs[j * D + c], o[c] * alpha);
        m = mn;
    }
    const float inv = 1.f / l;
    float* orow = out + ((size_t)b * 192 + t) * E + h * D;
#pragma unroll
    for (int c = 0; c < D; ++c) orow[c] = o[c] * inv;
}

// patch-embedding im2col with the hi / lo split of the f32 pixels
__global__ __launch_bounds__(256) void patch_im2col_split_kernel(const float* __restrict__ img, bf16* __restrict__ hi,
                                                                 bf16* __restrict__ lo, int B, int C, int H, int W,
                                                                 int Hp, int Wp, int p, int stride, int pad) {
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
        *reinterpret_cast<bf16x8*>(hi + tok * K + k) = oh;
        *reinterpret_cast<bf16x8*>(lo + tok * K + k) = ol;
    }
}

}  // namespace

extern "C" int vp_finish_f32(const float* x, int ldx, const float* colscale, const float* colshift, int act,
                             const float* resid, int ldr, const float* rowscale, int rows_per_scale, float* out,
                             void* hi, void* lo, int ldo, int M, int N, const int row_map[4], hipStream_t stream) {
    VP_CHECK_ARG(x && (out || hi) && M > 0 && N > 0 && N % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0);
    VP_CHECK_ARG(act >= 0 && act <= 2 && (!resid || ldr % 4 == 0) && (!rowscale || rows_per_scale > 0) && (!lo || hi));
    const int rd = row_map ? row_map[0] : 0;
    hipLaunchKernelGGL(finish_kernel, dim3(grid_for((size_t)M * (N / 4))), dim3(256), 0, stream, x, ldx, colscale,
                       colshift, act, resid, ldr, rowscale, rows_per_scale, out, (bf16*)hi, (bf16*)lo, ldo, M, N, rd,
                       rd ? row_map[1] : 0, rd ? row_map[2] : 0, rd ? row_map[3] : 0);
    VP_LAUNCH_RET();
}

extern "C" int vp_split_f32(const float* x, void* hi, void* lo, float* lo_f32, size_t n, hipStream_t stream) {
    VP_CHECK_ARG(x && (hi || lo || lo_f32) && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(split_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, x, (bf16*)hi, (bf16*)lo, lo_f32,
                       n / 4);
    VP_LAUNCH_RET();
}

extern "C" int vp_attention_f32(const float* qkv, float* out, int B, int H, int head_dim, int tokens, float scale,
                                hipStream_t stream) {
    VP_CHECK_ARG(qkv && out && B > 0 && H > 0);
    if (tokens != 192) return VP_ERR_UNSUPPORTED;
    const int E = H * head_dim;
    const size_t lds = 2 * 192 * (size_t)head_dim * sizeof(float);
    dim3 grid(B * H), block(192);
#define VP_AF(D)                                                                                              \
    (void)hipFuncSetAttribute((const void*)attn_f32_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(attn_f32_kernel<D>, grid, block, lds, stream, qkv, out, H, E, scale)
    switch (head_dim) {
        case 32: VP_AF(32); break;
        case 64: VP_AF(64); break;
        case 80: VP_AF(80); break;
        default: return VP_ERR_UNSUPPORTED;
    }
#undef VP_AF
    VP_LAUNCH_RET();
}

extern "C" int vp_patch_im2col_split(const float* img, void* col_hi, void* col_lo, int B, int C, int H, int W,
                                     int patch, int stride, int pad, hipStream_t stream) {
    VP_CHECK_ARG(img && col_hi && col_lo && B > 0 && C > 0 && patch % 8 == 0 && stride > 0);
    const int Hp = (H + 2 * pad - patch) / stride + 1, Wp = (W + 2 * pad - patch) / stride + 1;
    VP_CHECK_ARG(Hp > 0 && Wp > 0);
    const size_t total = (size_t)B * Hp * Wp * (C * patch * patch / 8);
    hipLaunchKernelGGL(patch_im2col_split_kernel, dim3(grid_for(total)), dim3(256), 0, stream, img, (bf16*)col_hi,
                       (bf16*)col_lo, B, C, H, W, Hp, Wp, patch, stride, pad);
    VP_LAUNCH_RET();
}
