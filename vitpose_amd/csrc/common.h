// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the ViTPose hot path.
// Wavefront = 64 lanes; MFMA fragments follow the v_mfma_f32_16x16x32_bf16 lane maps:
//   A operand: lane l holds A[row l&15][k = 8*(l>>4) + 0..7]
//   B operand: lane l holds B[k = 8*(l>>4) + 0..7][col l&15]
//   C/D:       lane l holds D[row 4*(l>>4) + r][col l&15], r = 0..3
#pragma once

// Clock probe builds.  -DVP_CLOCK_PROBE=1: lane 0 of workgroup 0 prints the shader-clock and the constant-rate (100 MHz)
// counter over a kernel's K loop, i.e. the core clock the loop actually ran at.  -DVP_CLOCK_PROBE=2: lane 0 of EVERY
// workgroup records (vp_tile_log, read back with vp_probe_dump) where it ran (XCC id, hardware id) and the constant-rate counter at its start, at the end of its K
// loop and after its last store has been acknowledged: the timeline of a launch, tile by tile.  Empty in the product build.
#if defined(VP_CLOCK_PROBE) && VP_CLOCK_PROBE == 2
#define VP_CLK_BEGIN() const long long vp_wall0 = wall_clock64(); long long vp_wallk = 0
#define VP_CLK_END(name) vp_wallk = wall_clock64()
#define VP_CLK_FINAL(name, tile)                                                                              \
    do {                                                                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        if (threadIdx.x == 0 && blockIdx.x < VP_TILE_LOG_MAX) {                                               \
            long long* vp_r = vp_tile_log + (size_t)blockIdx.x * 8;                                           \
            vp_r[0] = (long long)(tile);                                                                      \
            vp_r[1] = nk;                                                                                     \
            vp_r[2] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                                             \
            vp_r[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                                              \
            vp_r[4] = vp_wall0;                                                                               \
            vp_r[5] = vp_wallk;                                                                               \
            vp_r[6] = wall_clock64();                                                                         \
        }                                                                                                     \
    } while (0)
// generic per-workgroup time stamps (words 0..7 of the workgroup's record)
#define VP_TLOG(slot)                                                                                        \
    do {                                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < VP_TILE_LOG_MAX) vp_tile_log[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); \
    } while (0)
#elif defined(VP_CLOCK_PROBE)
#define VP_TLOG(slot) do {} while (0)
#define VP_CLK_BEGIN() const long long vp_clk0 = clock64(), vp_wall0 = wall_clock64()
#define VP_CLK_END(name)                                                                                      \
    do {                                                                                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0)                                                              \
            printf("clkprobe %s nk %d shader %lld wall %lld\n", name, nk, clock64() - vp_clk0, wall_clock64() - vp_wall0); \
    } while (0)
#define VP_CLK_FINAL(name, tile) do {} while (0)
#else
#define VP_TLOG(slot) do {} while (0)
#define VP_CLK_BEGIN() do {} while (0)
#define VP_CLK_END(name) do {} while (0)
#define VP_CLK_FINAL(name, tile) do {} while (0)
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(VP_CLOCK_PROBE) && VP_CLOCK_PROBE == 2
constexpr int VP_TILE_LOG_MAX = 8192;
static __device__ long long vp_tile_log[VP_TILE_LOG_MAX * 8];   // one record of eight words per workgroup of the last launch
#endif

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

// GELU and its derivative WITHOUT transcendental instructions (v_exp / v_rcp issue at a quarter of the VALU rate, and
// an erf evaluation with one v_rcp and one v_exp (Abramowitz & Stegun 7.1.26) costs ~90 VALU cycles per value: for a K = 768 product that is more VALU time in the epilogue
// than MFMA time in the K loop).  With a = min(|x|, 4.5) and s = 2 a / 4.5 - 1:
//   gelu(x)  = (x + |x|) / 2 - h(a),          h(a) = a erfc(a / sqrt 2) / 2
//   gelu'(x) = 1/2 + copysign(1/2 - d(a), x), d(a) = h'(a)
// h and d are smooth, decay like a Gaussian and are fitted by degree-11 polynomials in s (tools/fit_gelu_poly.py:
// max |error| 2.5e-5 for gelu and 7.5e-5 for gelu' over all x, the clamp included -- one to two orders below the
// bf16 resolution of the values they are stored as, and below what the bf16 rounding of the pre-activation itself
// does to them; degree 15 on [0, 5] reaches 2e-6 / 8e-6 for four more FMAs per value, which showed as 1-2 us per
// K = 768 launch: the arithmetic of these epilogues is 7-10 us of a 75-110 us launch, measured by removing it).
// 13 full-rate FMAs per value, packed two per instruction.  NaN propagates through gelu (x enters the result arithmetically).
// (four values at a time: the two packed FMAs of a Horner step are independent, so the dependent chain of one never
// waits on itself)
template <int NC>
__device__ __forceinline__ f32x4 gelu_poly(f32x4 s, const float (&c)[NC]) {
    f32x4 p = f32x4{c[0], c[0], c[0], c[0]};
#pragma unroll
    for (int i = 1; i < NC; ++i) p = __builtin_elementwise_fma(p, s, f32x4{c[i], c[i], c[i], c[i]});
    return p;
}
__device__ __forceinline__ f32x4 gelu_arg(f32x4 ax) {
    const f32x4 a = __builtin_elementwise_min(ax, f32x4{4.5f, 4.5f, 4.5f, 4.5f});
    constexpr float k = 2.0f / 4.5f;
    return __builtin_elementwise_fma(a, f32x4{k, k, k, k}, f32x4{-1.0f, -1.0f, -1.0f, -1.0f});
}
__device__ __forceinline__ f32x4 gelu_erf(f32x4 x) {
    constexpr float H[12] = {1.300883733e-02f, -4.582281338e-02f, -1.295990111e-02f, 1.839340585e-01f, -1.427471890e-01f,
                             -2.110001548e-01f, 4.206866235e-01f, -2.009975493e-01f, -1.448217540e-01f, 2.463749754e-01f,
                             -1.331519655e-01f, 2.750082184e-02f};
    const f32x4 ax = __builtin_elementwise_abs(x);
    const f32x4 h = gelu_poly(gelu_arg(ax), H);
    return __builtin_elementwise_fma(f32x4{0.5f, 0.5f, 0.5f, 0.5f}, x + ax, -h);
}
__device__ __forceinline__ f32x4 gelu_erf_grad(f32x4 x) {
    constexpr float D[12] = {6.466877457e-02f, 5.175437277e-02f, -3.583151178e-01f, -2.645623924e-02f, 7.865651507e-01f,
                             -4.628077661e-01f, -6.115162589e-01f, 9.403613011e-01f, -3.501445561e-01f, -1.936459407e-01f,
                             2.187125727e-01f, -5.917018482e-02f};
    const f32x4 d = gelu_poly(gelu_arg(__builtin_elementwise_abs(x)), D);
    const f32x4 half = f32x4{0.5f, 0.5f, 0.5f, 0.5f};
    return half + __builtin_elementwise_copysign(half - d, x);
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
