// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the ViTPose hot path.
// Wavefront = 64 lanes; MFMA fragments follow the v_mfma_f32_16x16x32_bf16 lane maps:
//   A operand: lane l holds A[row l&15][k = 8*(l>>4) + 0..7]
//   B operand: lane l holds B[k = 8*(l>>4) + 0..7][col l&15]
//   C/D:       lane l holds D[row 4*(l>>4) + r][col l&15], r = 0..3
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define VP_WAVE 64

#define VP_OK 0
#define VP_ERR_ARG (-1)     // bad argument (shape / alignment / null pointer)
#define VP_ERR_UNSUPPORTED (-2)

#define VP_CHECK_ARG(cond) do { if (!(cond)) return VP_ERR_ARG; } while (0)
#define VP_LAUNCH_RET() do { hipError_t e_ = hipGetLastError(); return e_ == hipSuccess ? VP_OK : (int)e_; } while (0)

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }  // round-to-nearest-even, NaN-preserving

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    bf16x4 r;
    r[0] = (bf16)a; r[1] = (bf16)b; r[2] = (bf16)c; r[3] = (bf16)d;
    return r;
}

__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Transposed LDS read: per 16-lane group a 4-row x 16-column block of 16-bit
// elements comes back column-major (lane i of the group gets column i of the 4 rows).
// Lane 4q+p of the group must pass the address of row q, columns 4p..4p+3.
__device__ __forceinline__ bf16x4 lds_read_tr(const char* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (bf16x4 __attribute__((address_space(3)))*)(lds_ptr));
}

// The same read issued as inline assembly, for kernels that fill LDS by LDS-DMA (global_load ... lds).  The
// compiler cannot prove that the builtin's read does not alias the LDS-DMA writes in flight, so it puts
// `s_waitcnt vmcnt(0)` in front of it: every outstanding prefetch -- including the ones issued a moment ago for a
// K tile two steps ahead -- has to land before the read issues, which removes the prefetch distance a multi-stage K
// loop is built around (those kernels order LDS-DMA against reads with their own counted waits + barriers).  The
// compiler does not see this read at all: it inserts no wait for its RESULT either, so every consumer must come
// after lds_fence() on the registers it reads.
// (`off`: a byte offset that is a compile-time constant after inlining; it travels in the instruction's offset field,
// so reads that differ only in it share one address register)
__device__ __forceinline__ bf16x4 lds_read_tr_raw(const char* lds_ptr, int off) {
#ifdef VP_TR_BUILTIN   /* A/B build: the builtin read and the compiler's own waits */
    return lds_read_tr(lds_ptr + off);
#else
    bf16x4 r;
    const unsigned a = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_ptr;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(off) : "memory");
    return r;
#endif
}
// Wait for every LDS read of this wave (lgkmcnt(0)); the registers passed in are tied to the wait, so that no
// instruction that reads them can be scheduled before it.
template <int N_>
__device__ __forceinline__ void lds_fence(bf16x8 (&f)[N_]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N_; ++i) asm volatile("" : "+v"(f[i]));
}
template <int N_, int S_>
__device__ __forceinline__ void lds_fence(bf16x8 (&f)[N_][S_]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N_; ++i)
#pragma unroll
        for (int j = 0; j < S_; ++j) asm volatile("" : "+v"(f[i][j]));
}

// erf(z) by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 resolution): one
// v_rcp, one v_exp and a 5-term Horner chain instead of the ~40-instruction libm erff, which made
// the GELU epilogues VALU-bound.  *e_out receives exp(-z^2) (shared with the Gaussian pdf).
__device__ __forceinline__ float erf_as(float z, float* e_out) {
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    const float e = __expf(-az * az);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    *e_out = e;
    return copysignf(1.0f - p * t * e, z);
}
__device__ __forceinline__ float gelu_erf(float x) {
    float e;
    return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f, &e));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float e;
    const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f, &e));
    return cdf + x * 0.39894228040143267794f * e;  // exp(-(x/sqrt2)^2) = exp(-x^2/2)
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (its L2),
// so give each of the 8 residue classes one contiguous chunk of the tile order.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
