// Fused global multi-head self-attention over the 192 tokens of a 256x192 crop
// (forward and backward), one 256-thread workgroup per (image, head).
//
// Input is the packed qkv activation [B*192][3E] (columns q | k | v, each head-major),
// exactly as the qkv Linear writes it; output is token-major [B*192][E].  No
// [B,h,N,d] permutes are materialised and the 192x192 score matrix never leaves
// registers: K/V (and Q/dO in backward) sit in LDS, each wave owns 16 query (or key)
// rows at a time and holds a full 192-wide score row, so softmax is single pass.
//
// MFMA orientation: scores are computed "key on the accumulator row, query on the
// lane" (D = K * Q^T), so the probabilities a lane holds are already the B operand of
// the P*V product (k order inside a 32-step = {4g+e, 16+4g+e}, matched by the
// transposed LDS reads of V); nothing crosses lanes except the 2-step row reduction.
#include "common.h"
#include "vitpose_hip.h"

namespace {

constexpr int NTOK = 192;

// One LDS image serves ds_read_b128 row reads and ds_read_b64_tr_b16 transposed
// reads: rows of D bf16, 32-byte units XOR-swizzled by a row key so that both the 8
// rows of a transposed read and the 16 rows of a row read fall on distinct banks.
// A reduction over the head dimension runs ceil(D/32) steps of the 16x16x32 MFMA.  When D % 32 == 16
// (head_dim 80, ViTPose-H) the last step carries 16 real columns in lane groups 0-1 and zeros in
// groups 2-3 of BOTH operands.  (A 16x16x16 MFMA for that tail, chained on the accumulator of a
// 16x16x32 MFMA issued just before it, returned stale values in half of the accumulator rows on
// gfx950 / ROCm 7.2; one opcode per accumulation chain avoids the mixed-shape dependency.)
template <int D>
struct Frag {
    static constexpr int NS = (D + 31) / 32;
    bf16x8 f[NS];
};

template <int D>
__device__ __forceinline__ f32x4 mfma_d(const Frag<D>& a, const Frag<D>& b, f32x4 acc) {
#pragma unroll
    for (int s = 0; s < Frag<D>::NS; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.f[s], b.f[s], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
    return z;
}

// 16 rows of a [rows][ld] bf16 matrix in global memory as an MFMA operand (lane -> row li, k = 8g..8g+7)
template <int D>
__device__ __forceinline__ Frag<D> frag_global(const bf16* row_li, int g) {
    Frag<D> r;
#pragma unroll
    for (int s = 0; s < D / 32; ++s) r.f[s] = *reinterpret_cast<const bf16x8*>(row_li + 32 * s + 8 * g);
    if constexpr (D % 32 == 16) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(row_li + (D - 16) + 8 * (g & 1));
        r.f[D / 32] = g < 2 ? v : zero8();
    }
    return r;
}

template <int D>
struct Img {
    // power-of-two head dims: rows of D bf16 with 32-byte units XOR-swizzled by a row key.
    // D = 80: 160-byte rows padded to a 176-byte pitch (44 dwords: 16 consecutive rows start on 16
    // distinct bank quads, so row reads are conflict-free without a swizzle).
    static constexpr bool POW2 = (D & (D - 1)) == 0;
    static constexpr int P = POW2 ? D * 2 : D * 2 + 16;   // row pitch in bytes
    static constexpr int U = D / 16;         // 32-byte units per row
    static constexpr int R = POW2 ? 256 / P : 1;          // rows per 256-byte bank row
    static constexpr int BYTES = NTOK * P;
    __device__ static __forceinline__ int key(int row) { return POW2 ? (row / R) % U : 0; }
    __device__ static __forceinline__ int off16(int row, int c16) {  // 16-byte chunk c16 of a row
        return row * P + ((((c16 >> 1) ^ key(row)) << 5) | ((c16 & 1) << 4));
    }
    // row-read fragment: rows [r0, r0+16), reduction step s (32 wide): lane -> row r0+li, chunk 4s+g
    __device__ static __forceinline__ bf16x8 rows(const char* img, int r0, int s, int lane) {
        return *reinterpret_cast<const bf16x8*>(img + off16(r0 + (lane & 15), 4 * s + (lane >> 4)));
    }
    // all reduction steps of rows [r0, r0+16)
    __device__ static __forceinline__ Frag<D> frag(const char* img, int r0, int lane) {
        Frag<D> r;
#pragma unroll
        for (int s = 0; s < D / 32; ++s) r.f[s] = rows(img, r0, s, lane);
        if constexpr (D % 32 == 16) {
            const int g = lane >> 4;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(img + off16(r0 + (lane & 15), (D - 16) / 8 + (g & 1)));
            r.f[D / 32] = g < 2 ? v : zero8();
        }
        return r;
    }
    // transposed fragment for the reduction over ROWS: 32-row step rs, 16-column unit dt.
    // element 4h+e of lane (g, li) = img[32 rs + 16 h + 4 g + e][16 dt + li]
    __device__ static __forceinline__ bf16x8 cols(const char* img, int rs, int dt, int lane) {
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int ra = 32 * rs + 4 * g + q, rb = ra + 16;
        bf16x4 lo = lds_read_tr(img + ra * P + ((dt ^ key(ra)) << 5) + (p << 3));
        bf16x4 hi = lds_read_tr(img + rb * P + ((dt ^ key(rb)) << 5) + (p << 3));
        return cat8(lo, hi);
    }
    // cooperative fill from a strided global matrix (row stride ld elements)
    __device__ static __forceinline__ void fill(char* img, const bf16* src, size_t ld, int tid) {
        constexpr int CH = D / 8;
        for (int idx = tid; idx < NTOK * CH; idx += 256) {
            const int row = idx / CH, c = idx % CH;
            *reinterpret_cast<uint4*>(img + off16(row, c)) =
                *reinterpret_cast<const uint4*>(src + (size_t)row * ld + c * 8);
        }
    }
};

// Workgroup -> (image, head).  Consecutive workgroup ids go to the eight XCDs in turn, each with its own L2.  A head's slice of
// a qkv row is head_dim * 2 bytes: 160 at head_dim 80, 64 at head_dim 32 -- not whole 128-byte lines, so neighbouring heads of
// one image share cache lines.  All heads of an image are therefore given to ONE XCD, back to back (ids congruent mod 8
// walk the heads of images xcd, xcd + 8, ...): the shared lines are fetched from memory once instead of once per head.
// Needs a batch that is a multiple of 8; otherwise the plain order.  Measured with cold operands: head_dim 80 at 128 images
// forward 122 -> 116 us, backward 262 -> 242 us; head_dim 32 at 64 images 22.7 -> 21.4 / 45.0 -> 43.5 us.
template <int D>
__device__ __forceinline__ void image_head(int id, int nblocks, int H, int& b, int& h) {
    const int B = nblocks / H;
    if ((D * 2) % 128 != 0 && (B & 7) == 0) {      // (whole-line slices, head_dim 64: nothing is shared, plain order)
        const int xcd = id & 7, i = id >> 3;
        b = (i / H) * 8 + xcd;
        h = i % H;
    } else {
        b = id / H;
        h = id % H;
    }
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = (bf16)a[i]; r[4 + i] = (bf16)b[i]; }
    return r;
}

template <int D>
__global__ __launch_bounds__(256, (D & (D - 1)) == 0 ? 3 : 2) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                          float* __restrict__ lse, int H, int E, float scale) {
    using I = Img<D>;
    __shared__ __attribute__((aligned(16))) char lds[2 * I::BYTES];
    char* kimg = lds;
    char* vimg = lds + I::BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    int b, h;
    image_head<D>(blockIdx.x, gridDim.x, H, b, h);
    const size_t ld = 3 * (size_t)E;
    const bf16* base = qkv + (size_t)b * NTOK * ld + h * D;
    I::fill(kimg, base + E, ld, tid);
    I::fill(vimg, base + 2 * E, ld, tid);
    __syncthreads();
    const float c = scale * 1.44269504088896340736f;
#pragma unroll 1
    for (int qt = 0; qt < 3; ++qt) {
        const int q0 = wave * 48 + qt * 16;
        const Frag<D> qf = frag_global<D>(base + (size_t)(q0 + li) * ld, g);
        f32x4 sc[12];
#pragma unroll
        for (int kt = 0; kt < 12; ++kt) {
            sc[kt] = mfma_d<D>(I::frag(kimg, 16 * kt, lane), qf, f32x4{0.f, 0.f, 0.f, 0.f});
            if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bounded look-ahead keeps the wave under 168 VGPRs
        }
        // sc[kt][r] = <q(q0+li), k(16kt+4g+r)>; softmax over keys = over kt, r and the 4 lane groups
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 12; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, sc[kt][r]);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
        const float mc = m * c;
#pragma unroll
        for (int kt = 0; kt < 12; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[kt][r] = __builtin_amdgcn_exp2f(sc[kt][r] * c - mc);
                l += sc[kt][r];
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (lse && g == 0) lse[(size_t)(b * H + h) * NTOK + q0 + li] = m * scale + __logf(l);
        const float inv = 1.f / l;
        f32x4 o[D / 16];
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const bf16x8 pb = pack8(sc[2 * ks], sc[2 * ks + 1]);
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::cols(vimg, ks, dt, lane), pb, o[dt], 0, 0, 0);
        }
        bf16* orow = out + ((size_t)b * NTOK + q0 + li) * E + h * D;
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt)
            *reinterpret_cast<bf16x4*>(orow + 16 * dt + 4 * g) =
                pack4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
    }
}

// Backward: recompute P from the forward's log-sum-exp.  Phase A (a wave owns 16 queries): S, dP -> dS -> dQ.
// Phase B (a wave owns 16 keys): S^T, dP^T -> P^T, dS^T -> dV, dK.  Every reduction is wave-local, and dq / dk / dv are
// written exactly once.  One 768-thread workgroup per (image, head) keeps the K, V, Q and dO images in LDS together
// (4 x 24 KiB at head_dim 64; one workgroup per CU), so every operand is read from memory ONCE, and runs both phases
// at the same time: waves 0-5 own two 16-query tiles each (phase A), waves 6-11 two 16-key tiles each (phase B).  One
// barrier in the whole kernel.  (A four-wave version with one PAIR of images that the two phases filled in turn --
// three workgroups per CU, K / V / Q fetched twice and dO three times -- ran 2-4 % slower at every head_dim.)  The
// kernel is bound by LDS fragment reads and exp2 latency, not by MFMA throughput or HBM.
template <int D>
__global__ __launch_bounds__(768, 1) void attn_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o_fwd,
                                                            const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                            bf16* __restrict__ dqkv, int H, int E, float scale) {
    using I = Img<D>;
    __shared__ __attribute__((aligned(16))) char lds[4 * I::BYTES + 2 * NTOK * 4];
    char* kimg = lds;
    char* vimg = lds + I::BYTES;
    char* qimg = lds + 2 * I::BYTES;
    char* dimg = lds + 3 * I::BYTES;
    float* s_lse = reinterpret_cast<float*>(lds + 4 * I::BYTES);
    float* s_del = s_lse + NTOK;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    VP_TLOG(0);
    int b, h;
    image_head<D>(blockIdx.x, gridDim.x, H, b, h);
    const size_t ld = 3 * (size_t)E;
    const bf16* base = qkv + (size_t)b * NTOK * ld + h * D;
    const bf16* dob = dout + (size_t)b * NTOK * E + h * D;
    const bf16* ofb = o_fwd + (size_t)b * NTOK * E + h * D;
    const float LOG2E = 1.44269504088896340736f;
    // fill the four images (every 16-byte chunk of the (image, head) slice is fetched once)
    {
        constexpr int CH = D / 8, N4 = NTOK * CH;
        for (int idx = tid; idx < N4; idx += 768) {
            const int row = idx / CH, c = idx % CH;
            const uint4 q = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + c * 8);
            const uint4 k = *reinterpret_cast<const uint4*>(base + E + (size_t)row * ld + c * 8);
            const uint4 v = *reinterpret_cast<const uint4*>(base + 2 * E + (size_t)row * ld + c * 8);
            const uint4 d = *reinterpret_cast<const uint4*>(dob + (size_t)row * E + c * 8);
            const int off = I::off16(row, c);
            *reinterpret_cast<uint4*>(qimg + off) = q;
            *reinterpret_cast<uint4*>(kimg + off) = k;
            *reinterpret_cast<uint4*>(vimg + off) = v;
            *reinterpret_cast<uint4*>(dimg + off) = d;
        }
    }
    // delta[q] = <dO[q], O[q]>: wave w takes rows 16 w .. 16 w + 15 as MFMA-layout fragments straight from memory
    // (the dO rows are in flight for the image anyway: a cache hit), one partial dot per lane, summed over the 4 lane
    // groups
    {
        const int row = wave * 16 + li;
        const Frag<D> df = frag_global<D>(dob + (size_t)row * E, g);
        const Frag<D> of = frag_global<D>(ofb + (size_t)row * E, g);
        float d = 0.f;
#pragma unroll
        for (int s = 0; s < Frag<D>::NS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) d += (float)df.f[s][j] * (float)of.f[s][j];
        d += __shfl_xor(d, 16, 64);
        d += __shfl_xor(d, 32, 64);
        if (g == 0) {
            s_del[row] = d;
            s_lse[row] = lse[(size_t)(b * H + h) * NTOK + row] * LOG2E;
        }
    }
    __syncthreads();
    VP_TLOG(1);
    const float c = scale * LOG2E;
    bf16* dq_base = dqkv + (size_t)b * NTOK * ld + h * D;

    if (wave < 6) {
        // ---------------- phase A: dQ of query tiles 2 wave, 2 wave + 1 ----------------
#pragma unroll 1
        for (int qt = 0; qt < 2; ++qt) {
            const int q0 = (wave * 2 + qt) * 16;
            const Frag<D> qf = I::frag(qimg, q0, lane);
            const Frag<D> df = I::frag(dimg, q0, lane);
            const float my_lse = s_lse[q0 + li], my_del = s_del[q0 + li];
            f32x4 ds[12];
#pragma unroll
            for (int kt = 0; kt < 12; ++kt) {
                const f32x4 sv = mfma_d<D>(I::frag(kimg, 16 * kt, lane), qf, f32x4{0.f, 0.f, 0.f, 0.f});
                const f32x4 pv = mfma_d<D>(I::frag(vimg, 16 * kt, lane), df, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[kt][r] = __builtin_amdgcn_exp2f(sv[r] * c - my_lse) * (pv[r] - my_del);
                if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bounded look-ahead (register budget)
            }
            f32x4 o[D / 16];
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const bf16x8 sb = pack8(ds[2 * ks], ds[2 * ks + 1]);
#pragma unroll
                for (int dt = 0; dt < D / 16; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::cols(kimg, ks, dt, lane), sb, o[dt], 0, 0, 0);
            }
            bf16* row = dq_base + (size_t)(q0 + li) * ld;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt)
                *reinterpret_cast<bf16x4*>(row + 16 * dt + 4 * g) =
                    pack4(o[dt][0] * scale, o[dt][1] * scale, o[dt][2] * scale, o[dt][3] * scale);
        }
    } else {
        // ---------------- phase B: dK, dV of key tiles 2 (wave - 6), 2 (wave - 6) + 1 ----------------
#pragma unroll 1
        for (int kt = 0; kt < 2; ++kt) {
            const int k0 = ((wave - 6) * 2 + kt) * 16;
            const Frag<D> kf = I::frag(kimg, k0, lane);
            const Frag<D> vf = I::frag(vimg, k0, lane);
            f32x4 dv[D / 16], dk[D / 16];
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) dv[dt] = dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int qs = 0; qs < 6; ++qs) {
                f32x4 pp[2], dd[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int qt = 2 * qs + hh;
                    const f32x4 sv = mfma_d<D>(I::frag(qimg, 16 * qt, lane), kf, f32x4{0.f, 0.f, 0.f, 0.f});
                    const f32x4 pv = mfma_d<D>(I::frag(dimg, 16 * qt, lane), vf, f32x4{0.f, 0.f, 0.f, 0.f});
                    // sv[r] = <q(16qt+4g+r), k(k0+li)>
                    const f32x4 ls = *reinterpret_cast<const f32x4*>(s_lse + 16 * qt + 4 * g);
                    const f32x4 dl = *reinterpret_cast<const f32x4*>(s_del + 16 * qt + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pp[hh][r] = __builtin_amdgcn_exp2f(sv[r] * c - ls[r]);
                        dd[hh][r] = pp[hh][r] * (pv[r] - dl[r]);
                    }
                }
                const bf16x8 pb = pack8(pp[0], pp[1]);
                const bf16x8 sb = pack8(dd[0], dd[1]);
#pragma unroll
                for (int dt = 0; dt < D / 16; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::cols(dimg, qs, dt, lane), pb, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::cols(qimg, qs, dt, lane), sb, dk[dt], 0, 0, 0);
                }
            }
            bf16* row = dq_base + (size_t)(k0 + li) * ld;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                *reinterpret_cast<bf16x4*>(row + E + 16 * dt + 4 * g) =
                    pack4(dk[dt][0] * scale, dk[dt][1] * scale, dk[dt][2] * scale, dk[dt][3] * scale);
                *reinterpret_cast<bf16x4*>(row + 2 * E + 16 * dt + 4 * g) = pack4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
            }
        }
    }
#if defined(VP_CLOCK_PROBE) && VP_CLOCK_PROBE == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (probe build: the stamp is taken when the wave's stores are acknowledged)
    if (wave == 0) VP_TLOG(2);
    if (threadIdx.x == 6 * 64 && blockIdx.x < VP_TILE_LOG_MAX) vp_tile_log[(size_t)blockIdx.x * 8 + 3] = wall_clock64();
#endif
}

// ---------------------------------------------------------------------------------------------------
// fp32-accurate forward (evaluation): f32 packed qkv in, hi / lo pair array out.  Q, K, V and the
// probabilities are split into hi = bf16(x), lo = bf16(x - hi) and every product runs as
// hi*hi + hi*lo + lo*hi on the bf16 matrix cores (f32 accumulation, f32 softmax).  K and V hi / lo images
// sit in LDS (four images: one workgroup per CU).
template <int D>
__device__ __forceinline__ void frag_global_split(const float* row_li, int g, Frag<D>& hi, Frag<D>& lo) {
    auto cvt = [](const float* p, bf16x8& h, bf16x8& l) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = (bf16)a[j];
            h[4 + j] = (bf16)b[j];
            l[j] = (bf16)(a[j] - (float)h[j]);
            l[4 + j] = (bf16)(b[j] - (float)h[4 + j]);
        }
    };
#pragma unroll
    for (int s = 0; s < D / 32; ++s) cvt(row_li + 32 * s + 8 * g, hi.f[s], lo.f[s]);
    if constexpr (D % 32 == 16) {
        bf16x8 h, l;
        cvt(row_li + (D - 16) + 8 * (g & 1), h, l);
        hi.f[D / 32] = g < 2 ? h : zero8();
        lo.f[D / 32] = g < 2 ? l : zero8();
    }
}

template <int D>
__global__ __launch_bounds__(256, 1) void attn_split_kernel(const float* __restrict__ qkv, bf16* __restrict__ out_pair,
                                                            int H, int E, float scale) {
    using I = Img<D>;
    __shared__ __attribute__((aligned(16))) char lds[4 * I::BYTES];
    char* kh = lds;
    char* kl = lds + I::BYTES;
    char* vh = lds + 2 * I::BYTES;
    char* vl = lds + 3 * I::BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    int b, h;
    image_head<D>(blockIdx.x, gridDim.x, H, b, h);
    const size_t ld = 3 * (size_t)E;
    const float* base = qkv + (size_t)b * NTOK * ld + h * D;
    constexpr int CH = D / 8;
    for (int idx = tid; idx < NTOK * CH; idx += 256) {
        const int row = idx / CH, c8 = idx % CH;
        const float* pk = base + E + (size_t)row * ld + c8 * 8;
        const float* pv = base + 2 * E + (size_t)row * ld + c8 * 8;
        bf16x8 hi, lo;
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(pk), k1 = *reinterpret_cast<const f32x4*>(pk + 4);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(pv), v1 = *reinterpret_cast<const f32x4*>(pv + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hi[j] = (bf16)k0[j];
            hi[4 + j] = (bf16)k1[j];
            lo[j] = (bf16)(k0[j] - (float)hi[j]);
            lo[4 + j] = (bf16)(k1[j] - (float)hi[4 + j]);
        }
        *reinterpret_cast<bf16x8*>(kh + I::off16(row, c8)) = hi;
        *reinterpret_cast<bf16x8*>(kl + I::off16(row, c8)) = lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hi[j] = (bf16)v0[j];
            hi[4 + j] = (bf16)v1[j];
            lo[j] = (bf16)(v0[j] - (float)hi[j]);
            lo[4 + j] = (bf16)(v1[j] - (float)hi[4 + j]);
        }
        *reinterpret_cast<bf16x8*>(vh + I::off16(row, c8)) = hi;
        *reinterpret_cast<bf16x8*>(vl + I::off16(row, c8)) = lo;
    }
    __syncthreads();
    const float c = scale * 1.44269504088896340736f;
#pragma unroll 1
    for (int qt = 0; qt < 3; ++qt) {
        const int q0 = wave * 48 + qt * 16;
        Frag<D> qh, ql;
        frag_global_split<D>(base + (size_t)(q0 + li) * ld, g, qh, ql);
        f32x4 sc[12];
#pragma unroll
        for (int kt = 0; kt < 12; ++kt) {
            const Frag<D> fh = I::frag(kh, 16 * kt, lane);
            f32x4 a = mfma_d<D>(fh, qh, f32x4{0.f, 0.f, 0.f, 0.f});
            a = mfma_d<D>(fh, ql, a);
            sc[kt] = mfma_d<D>(I::frag(kl, 16 * kt, lane), qh, a);
            if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 12; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, sc[kt][r]);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
        const float mc = m * c;
#pragma unroll
        for (int kt = 0; kt < 12; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[kt][r] = exp2f(sc[kt][r] * c - mc);
                l += sc[kt][r];
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.f / l;
        f32x4 o[D / 16];
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const bf16x8 ph = pack8(sc[2 * ks], sc[2 * ks + 1]);
            f32x4 r0, r1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r0[j] = sc[2 * ks][j] - (float)ph[j];
                r1[j] = sc[2 * ks + 1][j] - (float)ph[4 + j];
            }
            const bf16x8 pl = pack8(r0, r1);
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const bf16x8 fvh = I::cols(vh, ks, dt, lane);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fvh, ph, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fvh, pl, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(I::cols(vl, ks, dt, lane), ph, o[dt], 0, 0, 0);
            }
        }
        bf16* orow = out_pair + ((size_t)b * NTOK + q0 + li) * 2 * E;
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
            const int n = h * D + 16 * dt + 4 * g;
            const f32x4 v = o[dt] * inv;
            const bf16x4 hi = pack4(v[0], v[1], v[2], v[3]);
            bf16* p = orow + ((n >> 5) << 6) + (n & 31);
            *reinterpret_cast<bf16x4*>(p) = hi;
            *reinterpret_cast<bf16x4*>(p + 32) =
                pack4(v[0] - (float)hi[0], v[1] - (float)hi[1], v[2] - (float)hi[2], v[3] - (float)hi[3]);
        }
    }
}

}  // namespace

extern "C" int vp_attention_split(const float* qkv, void* out_pair, int B, int H, int head_dim, int tokens, float scale,
                                  hipStream_t stream) {
    VP_CHECK_ARG(qkv && out_pair && B > 0 && H > 0);
    if (tokens != NTOK) return VP_ERR_UNSUPPORTED;
    const int E = H * head_dim;
    VP_CHECK_ARG(E % 32 == 0);
    dim3 grid(B * H), block(256);
    switch (head_dim) {
        case 32: hipLaunchKernelGGL(attn_split_kernel<32>, grid, block, 0, stream, qkv, (bf16*)out_pair, H, E, scale); break;
        case 64: hipLaunchKernelGGL(attn_split_kernel<64>, grid, block, 0, stream, qkv, (bf16*)out_pair, H, E, scale); break;
        case 80: hipLaunchKernelGGL(attn_split_kernel<80>, grid, block, 0, stream, qkv, (bf16*)out_pair, H, E, scale); break;
        default: return VP_ERR_UNSUPPORTED;
    }
    VP_LAUNCH_RET();
}

extern "C" int vp_attention_fwd(const void* qkv, void* out, float* lse, int B, int H, int head_dim, int tokens,
                                float scale, hipStream_t stream) {
    VP_CHECK_ARG(qkv && out && B > 0 && H > 0);
    if (tokens != NTOK) return VP_ERR_UNSUPPORTED;
    const int E = H * head_dim;
    dim3 grid(B * H), block(256);
    switch (head_dim) {
        case 32: hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, block, 0, stream, (const bf16*)qkv, (bf16*)out, lse, H, E, scale); break;
        case 64: hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, block, 0, stream, (const bf16*)qkv, (bf16*)out, lse, H, E, scale); break;
        case 80: hipLaunchKernelGGL(attn_fwd_kernel<80>, grid, block, 0, stream, (const bf16*)qkv, (bf16*)out, lse, H, E, scale); break;
        default: return VP_ERR_UNSUPPORTED;
    }
    VP_LAUNCH_RET();
}

#if defined(VP_CLOCK_PROBE) && VP_CLOCK_PROBE == 2
// probe build only: the per-workgroup time stamps of the last instrumented attention launch
extern "C" int vp_probe_dump_attn(long long* host, int words) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vp_tile_log), (size_t)words * 8);
}
#endif

extern "C" int vp_attention_bwd(const void* qkv, const void* out_fwd, const void* dout, const float* lse, void* dqkv,
                                int B, int H, int head_dim, int tokens, float scale, hipStream_t stream) {
    VP_CHECK_ARG(qkv && out_fwd && dout && lse && dqkv && B > 0 && H > 0);
    if (tokens != NTOK) return VP_ERR_UNSUPPORTED;
    const int E = H * head_dim;
    dim3 grid(B * H), block(768);
    switch (head_dim) {
        case 32: hipLaunchKernelGGL(attn_bwd_kernel<32>, grid, block, 0, stream, (const bf16*)qkv, (const bf16*)out_fwd, (const bf16*)dout, lse, (bf16*)dqkv, H, E, scale); break;
        case 64: hipLaunchKernelGGL(attn_bwd_kernel<64>, grid, block, 0, stream, (const bf16*)qkv, (const bf16*)out_fwd, (const bf16*)dout, lse, (bf16*)dqkv, H, E, scale); break;
        case 80: hipLaunchKernelGGL(attn_bwd_kernel<80>, grid, block, 0, stream, (const bf16*)qkv, (const bf16*)out_fwd, (const bf16*)dout, lse, (bf16*)dqkv, H, E, scale); break;
        default: return VP_ERR_UNSUPPORTED;
    }
    VP_LAUNCH_RET();
}
