// bf16 MFMA GEMM with fused epilogues for gfx950.
//
//   C[i][j] = sum_k X[i][k] * Y[j][k]        i in [0,M), j in [0,N), k in [0,K)
//
// X ("row operand") and Y ("column operand") are bf16 and each may be stored
// reduction-contiguous (P[r*ld + k], "k-contiguous") or reduction-major
// (P[k*ld + r], "k-major").  That covers the three products of a Linear layer with
// one kernel body and no transposed copies:
//   forward  y = x W^T   : X = x  [M][K] k-contig, Y = W  [N][K] k-contig
//   dgrad    dx = dy W    : X = dy [M][N] k-contig, Y = W  [N][K] k-major
//   wgrad    dW = dy^T x  : X = dy [M][N] k-major,  Y = x  [M][K] k-major  (split-K into f32 slabs)
//
// Tiling: 128x128x64 per 256-thread workgroup, 2x2 waves of 64x64, each wave 4x4
// v_mfma_f32_16x16x32_bf16 tiles.  Operands go global -> LDS by LDS-DMA (the next tile's
// copy is issued before the MFMAs of the current one into the other LDS buffer; one
// barrier per K step).  LDS images are
// XOR-swizzled: k-contiguous tiles are read with ds_read_b128 (conflict-free),
// k-major tiles with ds_read_b64_tr_b16 (hardware transpose, conflict-free).
// The MFMA A operand is the Y tile and the B operand the X tile, so a lane ends
// up with 4 consecutive columns of one C row: 8/16-byte epilogue accesses.
#include <type_traits>

#include "common.h"
#include "vitpose_hip.h"

#ifdef VP_ABLATE_NOFOLD   /* A/B build: the kernels without the folded-LayerNorm epilogue paths */
#define VP_FOLD(x) false
#else
#define VP_FOLD(x) __builtin_expect(!!(x), 0)
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

// 32-byte-unit swizzle key of a k-major tile row (rows 0-3 / 8-11 of a 16-row
// group must land on 8 distinct 8-bank slots for the transposed read).
__device__ __forceinline__ int kmaj_key(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }

struct Operand {
    const bf16* p;
    int ld;
    int rows;  // extent of the non-reduction index
};

// ---- global -> LDS, asynchronously (LDS-DMA: global_load_lds_dwordx4) ----
// One wave instruction moves 64 x 16 B and lands them CONTIGUOUSLY in LDS (wave-uniform base +
// lane * 16), so the XOR swizzle of the LDS image is applied on the per-lane global SOURCE
// address instead: the lane whose data lands in 16-byte slot L of the tile fetches the chunk
// that the swizzled image keeps there (the swizzles are involutions).  No staging VGPRs and no
// ds_write instructions: the register->LDS write path (~79 B/clk/CU) was the bottleneck of the
// register-staged version, not the MFMAs.
template <bool KMAJ, int ROWS = BM>
__device__ __forceinline__ void stage_tile(const Operand& op, int r0, int k0, char* lds, int wave, int lane) {
    static_assert(!KMAJ || ROWS == 128, "the k-major image is laid out for 128 columns");
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int base = (i * 4 + wave) * 64;  // first 16-byte slot written by this wave instruction
        const int L = base + lane;
        const bf16* src;
        if constexpr (!KMAJ) {
            const int r = L >> 3, c = (L & 7) ^ (r & 7);
            int row = r0 + r;
            row = row < op.rows ? row : op.rows - 1;  // clamped rows only feed masked outputs
            src = op.p + (size_t)row * op.ld + k0 + c * 8;
        } else {
            const int kr = L >> 4, s16 = L & 15;
            const int unit = (s16 >> 1) ^ kmaj_key(kr);
            int col = r0 + (unit * 2 + (s16 & 1)) * 8;
            col = col <= op.rows - 8 ? col : op.rows - 8;
            src = op.p + (size_t)(k0 + kr) * op.ld + col;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + base * 16), 16, 0, 0);
    }
}

// ---- LDS -> MFMA fragment: rows [rbase, rbase+16) of the tile, k sub-step s (32 wide) ----
template <bool KMAJ>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int rbase, int s, int lane) {
    const int g = lane >> 4, li = lane & 15;
    if constexpr (!KMAJ) {
        const int r = rbase + li;
        return *reinterpret_cast<const bf16x8*>(lds + r * 128 + ((((s << 2) + g) ^ (r & 7)) << 4));
    } else {
        // k rows 32 s + 8 g + q and 4 below: the swizzle key depends on the lane part (8 g + q) only, so both rows of
        // both sub-steps read at constant offsets from one per-lane address
        const int q = li >> 2, p = li & 3, unit = rbase >> 4, kq = 8 * g + q;
        const char* base = lds + kq * 256 + ((unit ^ kmaj_key(kq)) << 5) + (p << 3);
        bf16x4 lo = lds_read_tr_raw(base, s * 8192);
        bf16x4 hi = lds_read_tr_raw(base, s * 8192 + 1024);
        return cat8(lo, hi);
    }
}

struct Tail {
    int full;   // workgroups [0, full) compute whole tiles
    int split;  // K slices per tail tile (<= 1: no tail handling)
    float* ws;  // [tail tiles * split][16 fragments][256 threads] f32x4 partials
};

struct Epi {
    int kind;
    void* out; int ldo;
    void* out2;
    const float* bias;
    const float* resid;
    const float* rowscale; int rows_per_scale;
    const void* aux; int ldaux;
    const float* pos; int pos_rows;
    const float* colscale;
    int row_div, row_hi, row_lo, row_off;  // out row = (i / div) * hi + (i % div) * lo + off when div > 0
    long split_stride;                     // elements between the K-slice slabs of VP_EPI_SPLITK_F32
    int pair_act;                          // VP_EPI_PAIR: 0 none, 1 exact GELU, 2 ReLU
    // LayerNorm fold (evaluation).  Consumer (VP_EPI_BF16 / GELU, ln_stats set): X holds the RAW residual stream in
    // bf16 and Y the weights scaled by gamma, so out = rstd (acc - mean s) + c with s = colscale, c = bias, and mean /
    // rstd of a row from its ln_slots pairs (sum, sum of squares) over the ln_inv_n^-1 channels.  Producer
    // (VP_EPI_RESID_F32, ln_stats_out set): besides the f32 stream, writes its bf16 copy to out2 and one (sum, sum
    // of squares) pair per row and column tile.
    const float* ln_stats;
    float* ln_stats_out;
    int ln_slots;
    float ln_inv_n, ln_eps;
};

// mean and rstd of the rows a lane holds (fragment row a = row row0 + 16 a + li), from the producer's per-tile sums.
// The four lanes that hold a row (one per lane group g) each fetch the slots j = g, g + 4, ... -- all loads of a call
// are issued before the first is used -- and the partial sums meet by two shuffles: a fixed order of additions, so a
// row's statistics do not depend on the batch around it.
constexpr int LN_MAX_SLOTS = 12;       // column tiles of the producer: 1280 channels on 128-wide tiles are 10

template <int NA>
__device__ __forceinline__ void ln_row_stats(const Epi& e, float (&mu)[NA], float (&rsd)[NA], int row0, int g, int li,
                                             int M) {
    float2 t[NA][LN_MAX_SLOTS / 4];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int row_raw = row0 + a * 16 + li;
        const int row = row_raw < M ? row_raw : M - 1;
        const float2* p = reinterpret_cast<const float2*>(e.ln_stats) + (size_t)row * e.ln_slots;
#pragma unroll
        for (int j = 0; j < LN_MAX_SLOTS / 4; ++j) {
            const int slot = g + 4 * j;
            t[a][j] = slot < e.ln_slots ? p[slot] : float2{0.f, 0.f};
        }
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        float sum = t[a][0].x, sq = t[a][0].y;
#pragma unroll
        for (int j = 1; j < LN_MAX_SLOTS / 4; ++j) {
            sum += t[a][j].x;
            sq += t[a][j].y;
        }
        sum += __shfl_xor(sum, 16, 64);
        sq += __shfl_xor(sq, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float m = sum * e.ln_inv_n;
        mu[a] = m;
        rsd[a] = rsqrtf(fmaxf(sq * e.ln_inv_n - m * m, 0.f) + e.ln_eps);
    }
}

// Epilogue of one wave's 64x64 sub-tile.  All per-column (bias / colscale) and per-row
// (DropPath scale, residual / position row pointer, remapped output row) operands are fetched
// before the 4x4 tile loop, and loads inside the loop use clamped in-bounds addresses, so the
// loop body is branch-free apart from the predicated store and its loads can be batched.
#ifdef VP_ABLATE_NOSTORE
#define VP_ST_OK(c) ((c) && e.ldo < 0)   /* ablation build: keeps the values live, never stores */
#else
#define VP_ST_OK(c) (c)
#endif

// The addend of the VP_EPI_RESID_F32 epilogue (residual row, else position row, else 0) of one wave's block, in the
// register layout of its accumulators: loaded BEFORE the K loop by kernels that own one output tile per CU, so that
// after the loop only the store half of the epilogue's HBM traffic is left.
template <int FM, int FN>
__device__ __forceinline__ void load_addend(const Epi& e, f32x4 (&pre)[FM][FN], int row0, int col0, int lane, int M,
                                            int N) {
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int a = 0; a < FM; ++a) {
        const int row_raw = row0 + a * 16 + li;
        const int row = row_raw < M ? row_raw : M - 1;
        const float* add = e.resid ? e.resid + (size_t)row * e.ldo
                                   : (e.pos ? e.pos + (size_t)(row % e.pos_rows) * e.ldo : nullptr);
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int c = col0 + b * 16 + g * 4;
            const int col = c + 3 < N ? c : N - 4;
            pre[a][b] = add ? *reinterpret_cast<const f32x4*>(add + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// LNF: the consumer side of a folded LayerNorm (VP_EPI_BF16 / GELU with Epi::ln_stats): a code path of its own, so
// that the ordinary epilogue keeps the registers (and the instruction stream) it has without the fold.
template <int EPI, int FM, int FN, bool LNF>
__device__ __forceinline__ void run_epilogue_impl(const Epi& e, f32x4 (&acc)[FM][FN], int row0, int col0, int lane,
                                                  int M, int N, int slice, const f32x4 (*pre)[FN]) {
    const int g = lane >> 4, li = lane & 15;
    int col[FN];
    bool cok[FN];
    f32x4 bias[FN], cs[FN];
#pragma unroll
    for (int b = 0; b < FN; ++b) {
        const int c = col0 + b * 16 + g * 4;
        cok[b] = c + 3 < N;
        col[b] = cok[b] ? c : N - 4;
        bias[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        cs[b] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    if constexpr (EPI != VP_EPI_SPLITK_F32 && EPI != VP_EPI_DGELU && EPI != VP_EPI_DRELU) {
        if (e.bias) {
#pragma unroll
            for (int b = 0; b < FN; ++b) bias[b] = *reinterpret_cast<const f32x4*>(e.bias + col[b]);
        }
    }
    if constexpr (EPI == VP_EPI_BF16 || EPI == VP_EPI_SCALE_RELU || EPI == VP_EPI_PAIR || LNF) {
        if (e.colscale) {
#pragma unroll
            for (int b = 0; b < FN; ++b) cs[b] = *reinterpret_cast<const f32x4*>(e.colscale + col[b]);
        }
    }
    float mu[LNF ? FM : 1], rsd[LNF ? FM : 1];
    if constexpr (LNF) ln_row_stats<FM>(e, mu, rsd, row0, g, li, M);
#pragma unroll
    for (int a = 0; a < FM; ++a) {
        const int row_raw = row0 + a * 16 + li;
        const bool rok = row_raw < M;
        const int row = rok ? row_raw : M - 1;
        const size_t orow = e.row_div > 0
                                ? (size_t)(row / e.row_div) * e.row_hi + (row % e.row_div) * e.row_lo + e.row_off
                                : (size_t)row;
        if constexpr (EPI == VP_EPI_BF16 || EPI == VP_EPI_GELU || EPI == VP_EPI_SCALE_RELU) {
            bf16* o = (bf16*)e.out + orow * e.ldo;
            bf16* o2 = EPI == VP_EPI_GELU && e.out2 ? (bf16*)e.out2 + orow * e.ldo : nullptr;
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                f32x4 v;
                if constexpr (LNF) v = (acc[a][b] - cs[b] * mu[a]) * rsd[a] + bias[b];
                else v = acc[a][b] * cs[b] + bias[b];
                const bool ok = rok && cok[b];
                if constexpr (EPI == VP_EPI_GELU) {
                    if (o2 && VP_ST_OK(ok)) *reinterpret_cast<bf16x4*>(o2 + col[b]) = pack4(v[0], v[1], v[2], v[3]);
                    v = gelu_erf(v);
                }
                if constexpr (EPI == VP_EPI_SCALE_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (VP_ST_OK(ok)) *reinterpret_cast<bf16x4*>(o + col[b]) = pack4(v[0], v[1], v[2], v[3]);
            }
        } else if constexpr (EPI == VP_EPI_PAIR) {
            // hi / lo pair layout: column n of the row lives at (n / 32) * 64 + n % 32 (hi) and 32 further on (lo)
            bf16* o = (bf16*)e.out + orow * e.ldo;
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                f32x4 v = acc[a][b] * cs[b] + bias[b];
                if (e.pair_act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752440f));
                } else if (e.pair_act == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                const bf16x4 h = pack4(v[0], v[1], v[2], v[3]);
                const bf16x4 l = pack4(v[0] - (float)h[0], v[1] - (float)h[1], v[2] - (float)h[2], v[3] - (float)h[3]);
                const int pc = ((col[b] >> 5) << 6) | (col[b] & 31);
                if (VP_ST_OK(rok && cok[b])) {
                    *reinterpret_cast<bf16x4*>(o + pc) = h;
                    *reinterpret_cast<bf16x4*>(o + pc + 32) = l;
                }
            }
        } else if constexpr (EPI == VP_EPI_RESID_F32) {
            // out = addend + rowscale[row / rows_per_scale] * (acc + bias), addend = resid row or pos row
            const float rs = e.rowscale ? e.rowscale[row / e.rows_per_scale] : 1.f;
            const float* add = e.resid ? e.resid + (size_t)row * e.ldo
                                       : (e.pos ? e.pos + (size_t)(row % e.pos_rows) * e.ldo : nullptr);
            float* o = (float*)e.out + orow * e.ldo;
            f32x4 ad[FN];
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                if (pre) ad[b] = pre[a][b];
                else ad[b] = add ? *reinterpret_cast<const f32x4*>(add + col[b]) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                const f32x4 v = ad[b] + (acc[a][b] + bias[b]) * rs;
                if (VP_ST_OK(rok && cok[b])) *reinterpret_cast<f32x4*>(o + col[b]) = v;
            }
        } else if constexpr (EPI == VP_EPI_DGELU || EPI == VP_EPI_DRELU) {
            const bf16* ax = (const bf16*)e.aux + (size_t)row * e.ldaux;
            bf16* o = (bf16*)e.out + orow * e.ldo;
            bf16x4 u[FN];
#pragma unroll
            for (int b = 0; b < FN; ++b) u[b] = *reinterpret_cast<const bf16x4*>(ax + col[b]);
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                f32x4 v = acc[a][b];
                if constexpr (EPI == VP_EPI_DGELU) {
                    v *= gelu_erf_grad(__builtin_convertvector(u[b], f32x4));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (float)u[b][r] > 0.f ? v[r] : 0.f;
                }
                if (VP_ST_OK(rok && cok[b])) *reinterpret_cast<bf16x4*>(o + col[b]) = pack4(v[0], v[1], v[2], v[3]);
            }
        } else {  // VP_EPI_F32, VP_EPI_SPLITK_F32 (slice blockIdx.y of the partial-sum slab)
            float* o = (float*)e.out + orow * e.ldo;
            if constexpr (EPI == VP_EPI_SPLITK_F32) o += (size_t)slice * e.split_stride;
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                const f32x4 v = acc[a][b] + bias[b];
                if (VP_ST_OK(rok && cok[b])) *reinterpret_cast<f32x4*>(o + col[b]) = v;
            }
        }
    }
}

template <int EPI, int FM, int FN>
__device__ __forceinline__ void run_epilogue(const Epi& e, f32x4 (&acc)[FM][FN], int row0, int col0, int lane,
                                             int M, int N, int slice = 0, const f32x4 (*pre)[FN] = nullptr) {
    if constexpr (EPI == VP_EPI_BF16 || EPI == VP_EPI_GELU) {
        if (VP_FOLD(e.ln_stats)) return run_epilogue_impl<EPI, FM, FN, true>(e, acc, row0, col0, lane, M, N, slice, pre);
    }
    run_epilogue_impl<EPI, FM, FN, false>(e, acc, row0, col0, lane, M, N, slice, pre);
}

// bf16 outputs: a lane of the accumulator layout owns 4 consecutive columns (8 bytes), and 8-byte
// row-scattered stores leave the chip at ~2.5 TB/s (store-issue bound).  The wave's 64 x 64 block is
// therefore transposed through a wave-private LDS region (row pitch 144 B: conflict-free 8-byte
// writes, 16-byte aligned reads) and written as 16 bytes per lane, 128 contiguous bytes per row.
// (FN = 3, 48 columns: pitch 112 B has the same properties.)  RG = row fragments staged per flush.
constexpr int STG_PITCH = 144;
constexpr int STG_BYTES = 64 * STG_PITCH;

// RAW: the values leave as they are (no bias / scale / activation): the bf16 copy of an f32 output.
template <int EPI, int FM, int FN, int RG, bool RAW, bool LNF>
__device__ __forceinline__ void run_epilogue_staged_impl(const Epi& e, f32x4 (&acc)[FM][FN], int row0, int col0, int lane,
                                                         int M, int N, char* stg) {
    constexpr int PITCH = FN * 32 + 16, CHUNKS = FN * 2, ROWS = RG * 16;
    static_assert(FM % RG == 0, "row fragments must split evenly into flush groups");
    static_assert(EPI == VP_EPI_BF16 || EPI == VP_EPI_GELU || EPI == VP_EPI_SCALE_RELU || EPI == VP_EPI_DGELU ||
                  EPI == VP_EPI_DRELU, "bf16 outputs only");
    const int g = lane >> 4, li = lane & 15;
    int col[FN];
    f32x4 bias[FN], cs[FN];
#pragma unroll
    for (int b = 0; b < FN; ++b) {
        const int c = col0 + b * 16 + g * 4;
        col[b] = c + 3 < N ? c : N - 4;
        bias[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        cs[b] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    if constexpr (!RAW && (EPI == VP_EPI_BF16 || EPI == VP_EPI_GELU || EPI == VP_EPI_SCALE_RELU)) {
        if (e.bias) {
#pragma unroll
            for (int b = 0; b < FN; ++b) bias[b] = *reinterpret_cast<const f32x4*>(e.bias + col[b]);
        }
    }
    if constexpr (!RAW && (EPI == VP_EPI_BF16 || EPI == VP_EPI_SCALE_RELU || LNF)) {
        if (e.colscale) {
#pragma unroll
            for (int b = 0; b < FN; ++b) cs[b] = *reinterpret_cast<const f32x4*>(e.colscale + col[b]);
        }
    }
    constexpr int NPASS = EPI == VP_EPI_GELU ? 2 : 1;   // GELU: pass 0 = pre-activation copy (out2), pass 1 = out
#pragma unroll
    for (int half = 0; half < FM / RG; ++half) {
        float mu[LNF ? RG : 1], rsd[LNF ? RG : 1];      // folded LayerNorm: statistics of this flush group's rows
        if constexpr (LNF) ln_row_stats<RG>(e, mu, rsd, row0 + half * RG * 16, g, li, M);
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (EPI == VP_EPI_GELU && pass == 0 && !e.out2) continue;
#pragma unroll
            for (int a = 0; a < RG; ++a) {
                const int A = half * RG + a;
                bf16x4 u[FN];
                if constexpr (EPI == VP_EPI_DGELU || EPI == VP_EPI_DRELU) {
                    const int row_raw = row0 + A * 16 + li;
                    const int row = row_raw < M ? row_raw : M - 1;
                    const bf16* ax = (const bf16*)e.aux + (size_t)row * e.ldaux;
#pragma unroll
                    for (int b = 0; b < FN; ++b) u[b] = *reinterpret_cast<const bf16x4*>(ax + col[b]);
                }
#pragma unroll
                for (int b = 0; b < FN; ++b) {
                    f32x4 v;
                    if constexpr (EPI == VP_EPI_DGELU || EPI == VP_EPI_DRELU) {
                        v = acc[A][b];
                        if constexpr (EPI == VP_EPI_DGELU) {
                            v *= gelu_erf_grad(__builtin_convertvector(u[b], f32x4));
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = (float)u[b][r] > 0.f ? v[r] : 0.f;
                        }
                    } else if constexpr (RAW) {
                        v = acc[A][b];
                    } else {
                        if constexpr (LNF) v = (acc[A][b] - cs[b] * mu[a]) * rsd[a] + bias[b];
                        else v = acc[A][b] * cs[b] + bias[b];
                        if (EPI == VP_EPI_GELU && pass == 1) {
                            v = gelu_erf(v);
                        }
                        if constexpr (EPI == VP_EPI_SCALE_RELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                        }
                    }
                    *reinterpret_cast<bf16x4*>(stg + (a * 16 + li) * PITCH + (b * 16 + g * 4) * 2) =
                        pack4(v[0], v[1], v[2], v[3]);
                }
            }
            bf16* dst = (EPI == VP_EPI_GELU && pass == 0) ? (bf16*)e.out2 : (bf16*)e.out;
#pragma unroll
            for (int i = 0; i < (ROWS * CHUNKS + 63) / 64; ++i) {
                const int id = i * 64 + lane, r = id / CHUNKS, c = id - r * CHUNKS;
                const int grow = row0 + half * ROWS + r, gcol = col0 + c * 8;
                const bool in = id < ROWS * CHUNKS;
                const uint4 val = *reinterpret_cast<const uint4*>(stg + (in ? r * PITCH + c * 16 : 0));
                if (VP_ST_OK(in && grow < M && gcol + 7 < N)) {
                    const size_t orow = e.row_div > 0 ? (size_t)(grow / e.row_div) * e.row_hi +
                                                            (grow % e.row_div) * e.row_lo + e.row_off
                                                      : (size_t)grow;
                    *reinterpret_cast<uint4*>(dst + orow * e.ldo + gcol) = val;
                }
            }
        }
    }
}

template <int EPI, int FM, int FN = 4, int RG = 4, bool RAW = false>
__device__ __forceinline__ void run_epilogue_staged(const Epi& e, f32x4 (&acc)[FM][FN], int row0, int col0, int lane,
                                                    int M, int N, char* stg) {
    if constexpr (!RAW && (EPI == VP_EPI_BF16 || EPI == VP_EPI_GELU)) {
        if (VP_FOLD(e.ln_stats)) return run_epilogue_staged_impl<EPI, FM, FN, RG, false, true>(e, acc, row0, col0, lane, M, N, stg);
    }
    run_epilogue_staged_impl<EPI, FM, FN, RG, RAW, false>(e, acc, row0, col0, lane, M, N, stg);
}

// VP_EPI_RESID_F32 with the LayerNorm fold's by-products (kernels whose four column waves `wc` of a row group `grp`
// share their rows): the f32 stream as usual, then per row the sum and the sum of squares of the tile's columns --
// lane sums, the four lane groups, then the four waves through LDS in a fixed order -- stored once per (row, column
// tile), and the bf16 copy of the stream through the staging patches (16-byte stores).  `lds` must be free (a barrier
// behind the K loop) and hold >= 96 KiB + 8 KiB.
constexpr int LN_RED_OFF = 96 * 1024;

template <int FM, int FN, int RG>
__device__ __forceinline__ void resid_stats_epilogue(const Epi& e, f32x4 (&acc)[FM][FN], const f32x4 (*pre)[FN], int row0,
                                                     int col0, int lane, int M, int N, char* lds, int wave, int grp,
                                                     int wc, int tile_n, int stg_bytes) {
    const int g = lane >> 4, li = lane & 15;
    float* red = reinterpret_cast<float*>(lds + LN_RED_OFF);        // [2 groups][4 waves][FM * 16 rows][2]
    f32x4 bias[FN];
    int col[FN];
    bool cok[FN];
#pragma unroll
    for (int b = 0; b < FN; ++b) {
        const int c = col0 + b * 16 + g * 4;
        cok[b] = c + 3 < N;
        col[b] = cok[b] ? c : N - 4;
        bias[b] = e.bias ? *reinterpret_cast<const f32x4*>(e.bias + col[b]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int a = 0; a < FM; ++a) {
        const int row_raw = row0 + a * 16 + li;
        const bool rok = row_raw < M;
        const int row = rok ? row_raw : M - 1;
        const float rs = e.rowscale ? e.rowscale[row / e.rows_per_scale] : 1.f;
        const float* add = e.resid ? e.resid + (size_t)row * e.ldo
                                   : (e.pos ? e.pos + (size_t)(row % e.pos_rows) * e.ldo : nullptr);
        float* o = (float*)e.out + (size_t)row * e.ldo;
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            f32x4 ad;
            if (pre) ad = pre[a][b];
            else ad = add ? *reinterpret_cast<const f32x4*>(add + col[b]) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 v = ad + (acc[a][b] + bias[b]) * rs;
            acc[a][b] = v;
            if (VP_ST_OK(rok && cok[b])) *reinterpret_cast<f32x4*>(o + col[b]) = v;
            if (cok[b]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sum += v[r];
                    sq = __builtin_fmaf(v[r], v[r], sq);
                }
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sq += __shfl_xor(sq, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        if (g == 0) {
            float* r = red + (((grp * 4 + wc) * FM * 16) + a * 16 + li) * 2;
            r[0] = sum;
            r[1] = sq;
        }
    }
    __builtin_amdgcn_s_barrier();
    {
        const int r = wc * 64 + lane;            // the four waves of a row group share its FM * 16 rows
        if (r < FM * 16 && row0 + r < M) {
            float sum = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float* p = red + (((grp * 4 + w) * FM * 16) + r) * 2;
                sum += p[0];
                sq += p[1];
            }
            float2* dst = reinterpret_cast<float2*>(e.ln_stats_out) + (size_t)(row0 + r) * e.ln_slots + tile_n;
            *dst = float2{sum, sq};
        }
    }
    Epi e2 = e;
    e2.out = e.out2;
    e2.row_div = 0;
    run_epilogue_staged<VP_EPI_BF16, FM, FN, RG, true>(e2, acc, row0, col0, lane, M, N, lds + wave * stg_bytes);
}

template <int EPI>
constexpr bool epi_is_bf16() {
    return EPI == VP_EPI_BF16 || EPI == VP_EPI_GELU || EPI == VP_EPI_SCALE_RELU || EPI == VP_EPI_DGELU ||
           EPI == VP_EPI_DRELU;
}

// SPLIT3 (every tile schedule below): both operands are hi / lo PAIR arrays -- each 64-element K step of a row holds
// hi(x[32 j .. 32 j + 31]) then lo(the same 32 values), hi = bf16(x), lo = bf16(x - hi) -- and a K step
// accumulates hi*hi + hi*lo + lo*hi (three MFMAs per fragment pair, one accumulator set, one epilogue): an
// fp32-accurate product (the dropped lo*lo term is ~2^-16 relative) on the bf16 matrix cores.  The LDS images and
// loaders are those of the plain kernels: sub-step 0 of a staged tile is the hi half, sub-step 1 the lo half.
template <bool XKMAJ, bool YKMAJ, int EPI, bool SPLIT3 = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(Operand X, Operand Y, int M, int N, int K,
                                                           int tiles_n, int ntiles, int ksplit_tiles, Epi e,
                                                           Tail tl) {
    __shared__ __attribute__((aligned(16))) char lds[2][2][TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // Wave quantisation: the chip holds `slots` workgroups at once, so the last ntiles % slots tiles
    // would run as a mostly empty round.  Those tail tiles are instead cut along K into tl.split
    // slices (one workgroup each, all resident together); their f32 partials go to tl.ws and
    // gemm_tail_fixup sums them and applies the epilogue.
    int vb = blockIdx.x, tz = 0;
    const bool tail = tl.split > 1 && (int)blockIdx.x >= tl.full;
    if (tail) {
        const int j = blockIdx.x - tl.full;
        vb = tl.full + j / tl.split;
        tz = j - (j / tl.split) * tl.split;
    }

    // tile order: the N tiles of one 128-row panel are consecutive, and each XCD
    // (blocks congruent mod 8) gets a contiguous chunk of that order
    // ... and inside it tiles are walked in groups of 8 row panels, column by column, so the ~64
    // workgroups an XCD runs at once cover an 8x8 patch of tiles: 16 operand panels shared
    // through that XCD's 4 MiB L2 instead of a few row panels against the whole weight matrix
    const int t = xcd_remap(vb, ntiles);
    const int tiles_m = ntiles / tiles_n;
    const int gsz = 8 * tiles_n, gid = t / gsz, tin = t - gid * gsz;
    const int gm = min(tiles_m - gid * 8, 8);
    const int i0 = (gid * 8 + tin % gm) * BM, j0 = (tin / gm) * BN;
    const int nk_all = K / BK;
    const int kt0 = tail ? (tz * nk_all) / tl.split : blockIdx.y * ksplit_tiles;
    const int kt1 = tail ? ((tz + 1) * nk_all) / tl.split : min(nk_all, kt0 + ksplit_tiles);

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (kt0 < kt1) {  // an empty K slice still writes its (zero) partial tile
        stage_tile<XKMAJ>(X, i0, kt0 * BK, lds[0][0], wave, lane);
        stage_tile<YKMAJ>(Y, j0, kt0 * BK, lds[0][1], wave, lane);
        __syncthreads();  // (waits for the LDS-DMA: hipcc drains vmcnt before the barrier)

        for (int kt = kt0; kt < kt1; ++kt) {
            const int cur = (kt - kt0) & 1;
            // start the next tile's LDS-DMA into the other buffer (every wave finished reading it
            // before the barrier that ended the previous step), then run this tile's MFMAs under it
#ifndef VP_ABLATE_NOLOAD
            if (kt + 1 < kt1) {
                stage_tile<XKMAJ>(X, i0, (kt + 1) * BK, lds[cur ^ 1][0], wave, lane);
                stage_tile<YKMAJ>(Y, j0, (kt + 1) * BK, lds[cur ^ 1][1], wave, lane);
            }
#endif
            const char* lx = lds[cur][0];
            const char* ly = lds[cur][1];
            if constexpr (SPLIT3) {
                bf16x8 xf[4][2], yf[4][2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) xf[a][s] = read_frag<XKMAJ>(lx, wm * 64 + a * 16, s, lane);
#pragma unroll
                    for (int b = 0; b < 4; ++b) yf[b][s] = read_frag<YKMAJ>(ly, wn * 64 + b * 16, s, lane);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[b][t == 2], xf[a][t == 1], acc[a][b], 0, 0, 0);
            } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 xf[4], yf[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) xf[a] = read_frag<XKMAJ>(lx, wm * 64 + a * 16, s, lane);
#pragma unroll
                for (int b = 0; b < 4; ++b) yf[b] = read_frag<YKMAJ>(ly, wn * 64 + b * 16, s, lane);
                if constexpr (XKMAJ) lds_fence(xf);
                if constexpr (YKMAJ) lds_fence(yf);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
#ifndef VP_ABLATE_NOMFMA
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[b], xf[a], acc[a][b], 0, 0, 0);
#else
                        asm volatile("" ::"v"(yf[b]), "v"(xf[a]));
#endif
                    }
            }
            }
            __syncthreads();
        }
    }
    if (tail) {
        f32x4* w = reinterpret_cast<f32x4*>(tl.ws) + (size_t)(blockIdx.x - tl.full) * 16 * NTHREADS + tid;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) w[(a * 4 + b) * NTHREADS] = acc[a][b];
        return;
    }
    if constexpr (epi_is_bf16<EPI>()) {
        if (((N | e.ldo) & 7) == 0) {   // (the loop's last barrier: every wave is done with the operand tiles)
            run_epilogue_staged<EPI, 4>(e, acc, i0 + wm * 64, j0 + wn * 64, lane, M, N,
                                        &lds[0][0][0] + wave * STG_BYTES);
            return;
        }
    }
    run_epilogue<EPI, 4, 4>(e, acc, i0 + wm * 64, j0 + wn * 64, lane, M, N, blockIdx.y);
}

// One workgroup per tail tile: sum the K-slice partials (same thread <-> fragment mapping as
// the GEMM kernel) and run the fused epilogue.
template <int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_tail_fixup(int M, int N, int tiles_n, int ntiles, Epi e, Tail tl) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int t = xcd_remap(tl.full + blockIdx.x, ntiles);
    const int tiles_m = ntiles / tiles_n;
    const int gsz = 8 * tiles_n, gid = t / gsz, tin = t - gid * gsz;
    const int gm = min(tiles_m - gid * 8, 8);
    const int i0 = (gid * 8 + tin % gm) * BM, j0 = (tin / gm) * BN;
    f32x4 acc[4][4];
    const f32x4* w = reinterpret_cast<const f32x4*>(tl.ws) + (size_t)blockIdx.x * tl.split * 16 * NTHREADS + tid;
    // few workgroups run here, so keep many loads in flight per thread: all (b, z) partials of
    // one fragment row are fetched together (split <= 8; absent slices read slice 0 and are masked)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        f32x4 part[4][8];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int z = 0; z < 8; ++z)
                part[b][z] = w[((size_t)(z < tl.split ? z : 0) * 16 + a * 4 + b) * NTHREADS];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4 s = part[b][0];
#pragma unroll
            for (int z = 1; z < 8; ++z)
                if (z < tl.split) s += part[b][z];
            acc[a][b] = s;
        }
    }
    run_epilogue<EPI, 4, 4>(e, acc, i0 + wm * 64, j0 + wn * 64, lane, M, N, blockIdx.y);
}

#define VP_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// =====================================================================================
// 192 x 128 x 64 tile, 256 threads, TWO workgroups per CU
// =====================================================================================
// The one-workgroup-per-CU kernels below leave the MFMA pipes idle while their tile's epilogue drains
// (all CUs store at once: an HBM-write burst of ~13 us per launch, a quarter of a K = 768 GEMM).  Here
// two independent workgroups share a CU (2 x 80 KiB LDS, 2 waves / SIMD): one's epilogue and LDS reads
// run under the other's MFMAs, and after the first round the workgroups of a CU are out of step for
// good.  2 x 2 waves of 96 x 64 (6 x 4 MFMA tiles), two 40 KiB stages, one barrier per K step; X must
// be k-contiguous (forward and dgrad products).  Its main loop is slower than the ping-pong kernels'
// (the prefetch is only one K step deep), so it wins only where the epilogue is heavy: the exact-GELU
// epilogue with its second output.  (A variant with 32-wide K stages in a ring of four prefetched deeper
// but fetched half cache lines per row and was 10 % slower still.)
constexpr int W_BM = 192, W_BN = 128;
constexpr int W_XBYTES = W_BM * BK * 2, W_YBYTES = W_BN * BK * 2;   // 24 KiB + 16 KiB per stage

// Implicit sub-pixel im2col of a ConvTranspose2d(k=4, s=2, p=1): the X operand of output parity (py, px) is
//   X[(b, y, x)][t * C + c] = in[b][y + dy(py, ty)][x + dx(px, tx)][c]   (0 outside), t = ty * 2 + tx,
//   dy(0, {0, 1}) = {0, -1}, dy(1, {0, 1}) = {+1, 0}
// and is never written to memory: the LDS-DMA source address of a tile row is the shifted pixel's channel
// vector, or a page of zeros.  A 64-wide K step lies inside one tap (C % 64 == 0).
//
// taps16 (the data gradient of the same layer): rows are INPUT pixels of the H x W grid and
//   X[(b, iy, ix)][(ky * 4 + kx) * C + c] = dout[b][2 iy - 1 + ky][2 ix - 1 + kx][c]   (0 outside the 2H x 2W grid),
// the 16-tap neighbourhood of the OUTPUT-grid gradient that pixel (iy, ix) fed; one launch, no parities.
struct Gather {
    int H, W, C;
    const bf16* zero;   // >= 128 bytes of zeros
    int taps16;
};

struct GatherRows {      // per lane: the 6 tile rows it fetches (one per LDS-DMA round)
    int pix[6];          // pixel index of the row (clamped to the last row of the operand)
    int ok[6];           // bit t: tap t of this pixel lies inside the image
};

__device__ __forceinline__ GatherRows gather_rows(const Gather& g, int r0, int rows, int py, int px, int wave,
                                                  int lane) {
    GatherRows gr;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int L = (i * 4 + wave) * 64 + lane;
        int row = r0 + (L >> 3);
        row = row < rows ? row : rows - 1;
        const int x = row % g.W, y = (row / g.W) % g.H;
        int ok = 0;
        if (g.taps16) {
            const int b = row / (g.W * g.H);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int oy = 2 * y - 1 + (t >> 2), ox = 2 * x - 1 + (t & 3);
                ok |= (oy >= 0 && oy < 2 * g.H && ox >= 0 && ox < 2 * g.W) << t;
            }
            gr.pix[i] = (b * 2 * g.H + 2 * y) * 2 * g.W + 2 * x;     // output-grid pixel (2 iy, 2 ix)
            gr.ok[i] = ok;
            continue;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ty = t >> 1, tx = t & 1;
            const int yy = y + (py ? 1 - ty : -ty), xx = x + (px ? 1 - tx : -tx);
            ok |= (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) << t;
        }
        gr.pix[i] = row;
        gr.ok[i] = ok;
    }
    return gr;
}

__device__ __forceinline__ void stage_tile_gather(const bf16* in, const Gather& g, const GatherRows& gr, int k0,
                                                  int py, int px, char* lds, int wave, int lane) {
    const int t = k0 / g.C, c0 = k0 - t * g.C;            // wave-uniform
    const int ty = t >> 1, tx = t & 1;
    const int shift = g.taps16 ? ((t >> 2) - 1) * 2 * g.W + ((t & 3) - 1)
                               : (py ? 1 - ty : -ty) * g.W + (px ? 1 - tx : -tx);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int base = (i * 4 + wave) * 64, L = base + lane;
        const int r = L >> 3, c = (L & 7) ^ (r & 7);
        const bf16* src = ((gr.ok[i] >> t) & 1) ? in + (size_t)(gr.pix[i] + shift) * g.C + c0 + c * 8 : g.zero + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + base * 16), 16, 0, 0);
    }
}

template <bool YKMAJ, int EPI, bool GATHER = false, bool SPLIT3 = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_w2_kernel(Operand X, Operand Y, int M, int N, int K, int tiles_n,
                                                              int ntiles, Epi e, Gather gth = Gather{}) {
    __shared__ __attribute__((aligned(16))) char lds[2][W_XBYTES + W_YBYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int t = xcd_remap(blockIdx.x, ntiles);
    const int tiles_m = ntiles / tiles_n;
    const int gsz = 8 * tiles_n, gid = t / gsz, tin = t - gid * gsz;
    const int gm = min(tiles_m - gid * 8, 8);
    const int i0 = (gid * 8 + tin % gm) * W_BM, j0 = (tin / gm) * W_BN;
    const int nk = K / BK;

    f32x4 acc[6][4];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // GATHER: blockIdx.y = output parity (py, px) of the transposed conv; Y and the output rows follow it
    int py = 0, px = 0;
    GatherRows gr;
    if constexpr (GATHER) {
        if (!gth.taps16) {
            py = blockIdx.y >> 1;
            px = blockIdx.y & 1;
            Y.p += (size_t)blockIdx.y * Y.rows * Y.ld;
            e.row_div = gth.W;
            e.row_hi = 4 * gth.W;
            e.row_lo = 2;
            e.row_off = py * 2 * gth.W + px;
        }
        gr = gather_rows(gth, i0, M, py, px, wave, lane);
    }
    auto stage_x = [&](int k0, char* dst) {
        if constexpr (GATHER) stage_tile_gather(X.p, gth, gr, k0, py, px, dst, wave, lane);
        else stage_tile<false, W_BM>(X, i0, k0, dst, wave, lane);
    };
    stage_x(0, lds[0]);
    stage_tile<YKMAJ>(Y, j0, 0, lds[0] + W_XBYTES, wave, lane);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            stage_x((kt + 1) * BK, lds[cur ^ 1]);
            stage_tile<YKMAJ>(Y, j0, (kt + 1) * BK, lds[cur ^ 1] + W_XBYTES, wave, lane);
        }
        const char* lx = lds[cur];
        const char* ly = lds[cur] + W_XBYTES;
        if constexpr (SPLIT3) {
            bf16x8 xf[6][2], yf[4][2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int b = 0; b < 4; ++b) yf[b][s] = read_frag<YKMAJ>(ly, wn * 64 + b * 16, s, lane);
#pragma unroll
                for (int a = 0; a < 6; ++a) xf[a][s] = read_frag<false>(lx, wm * 96 + a * 16, s, lane);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[b][t == 2], xf[a][t == 1], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        } else {
            // the fragments of both 32-wide sub-steps are requested before the first MFMA: one LDS latency per K
            // tile instead of two (two waves per SIMD leave 256 VGPRs per wave)
            bf16x8 xf[6][2], yf[4][2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int b = 0; b < 4; ++b) yf[b][s] = read_frag<YKMAJ>(ly, wn * 64 + b * 16, s, lane);
#pragma unroll
                for (int a = 0; a < 6; ++a) xf[a][s] = read_frag<false>(lx, wm * 96 + a * 16, s, lane);
            }
            if constexpr (YKMAJ) lds_fence(yf);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[b][s], xf[a][s], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }
    if constexpr (epi_is_bf16<EPI>()) {
        if (((N | e.ldo) & 7) == 0) {   // the loop's last barrier: every wave is done with the operand tiles
            run_epilogue_staged<EPI, 6, 4, 3>(e, acc, i0 + wm * 96, j0 + wn * 64, lane, M, N,
                                              &lds[0][0] + wave * (48 * STG_PITCH));
            return;
        }
    }
    run_epilogue<EPI, 6, 4>(e, acc, i0 + wm * 96, j0 + wn * 64, lane, M, N);
}

// =====================================================================================
// 256 x 256 x 64 tile, 512 threads, one workgroup per CU ("ping-pong" schedule)
// =====================================================================================
// Waves 0-3 (group 0) own C rows [0,128) of the tile, waves 4-7 (group 1) rows [128,256); wave
// (g, wc) owns the 128 x 64 block at columns 64*wc: 8 x 4 MFMA fragments, computed per K tile
// as four 64 x 32 quadrants of 16 MFMAs.  Each quadrant is a phase  [R: LDS reads + LDS-DMA
// issue | barrier | C: 16 MFMAs | barrier];  group 1 runs ONE barrier behind group 0, so in
// every barrier interval one wave of each SIMD streams MFMAs while its partner reads LDS and
// issues loads -- the matrix pipe never waits for LDS, and no fragment is double-buffered.
//
// LDS: 2 stages x 4 half tiles (X rows 0-127, X rows 128-255, Y cols 0-127, Y cols 128-255;
// 16 KiB each, images as in the 128 x 128 kernel).  Every half tile is filled by all 512
// threads (2 LDS-DMA instructions each).  Schedule for K tile u (R(u,q) = read part of phase q):
//   R(u,0) reads X rows-quadrant 0 + Y column-quadrant 0, issues Y1(u+1)
//   R(u,1) reads Y column-quadrant 1,                      issues X0(u+1)
//   R(u,2) reads X rows-quadrant 1,                        issues X1(u+1)
//   R(u,3) reads nothing,                                  issues Y0(u+2)   (Y halves are free after R(u,1))
// so a slot is overwritten at least two barrier intervals after its last read and every load
// has at least three intervals to land.  Two counted waits per K tile make the data visible:
// vmcnt(4) at the end of R(u,3) (all but {X1(u+1), Y0(u+2)} landed: group 0 reads X0/Y(u+1) in
// the next interval) and vmcnt(2) at the end of C(u,3) (X1(u+1) landed: group 1 reads it two
// intervals later); each is followed by the barrier that every reader passes first.  Loads
// beyond the K range are still issued (clamped source, dead slot) so the counts stay exact.
constexpr int T2 = 256, T2_THREADS = 512, HALF_BYTES = 128 * BK * 2;

template <bool KMAJ>
__device__ __forceinline__ void stage_half(const Operand& op, int r0, int k0, char* lds, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int base = (i * 8 + wave) * 64;
        const int L = base + lane;
        const bf16* src;
        if constexpr (!KMAJ) {
            const int r = L >> 3, c = (L & 7) ^ (r & 7);
            int row = r0 + r;
            row = row < op.rows ? row : op.rows - 1;
            src = op.p + (size_t)row * op.ld + k0 + c * 8;
        } else {
            const int kr = L >> 4, s16 = L & 15;
            const int unit = (s16 >> 1) ^ kmaj_key(kr);
            int col = r0 + (unit * 2 + (s16 & 1)) * 8;
            col = col <= op.rows - 8 ? col : op.rows - 8;
            src = op.p + (size_t)(k0 + kr) * op.ld + col;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + base * 16), 16, 0, 0);
    }
}

// Grouped launch (several independent weight-gradient problems with one reduction extent in one grid):
// work id -> problem by a ballot over the table's first-tile column (<= 64 problems), then the problem's
// operands replace the kernel arguments (wave-uniform: read through the scalar path).
__device__ __forceinline__ float* select_problem(const vp_group_entry* grp, int ngrp, int lane, int& work, Operand& X,
                                                 Operand& Y, int& M, int& N, int& tiles_n, int& ntiles, Epi& e,
                                                 int* pidx = nullptr) {
    const int first = lane < ngrp ? grp[lane].tile0 : 0x7fffffff;
    const int p = __builtin_amdgcn_readfirstlane(__popcll(__ballot(first <= work)) - 1);
    if (pidx) *pidx = p;
    const vp_group_entry g = grp[p];
    X = Operand{(const bf16*)g.x, g.ldx, g.M};
    Y = Operand{(const bf16*)g.y, g.ldy, g.N};
    M = g.M;
    N = g.N;
    tiles_n = g.tiles_n;
    ntiles = g.ntiles;
    e.out = g.out;
    e.ldo = g.ldo;
    work -= g.tile0;
    return g.rowsum;
}

// Row sums of a grouped weight gradient, rowsum[m] = sum_k x[k][m] (the bias gradient that belongs to the
// weight gradient x^T y): in the first column tile of a row panel each X fragment gets one more MFMA against
// an all-ones operand (spread over the four waves that share the fragment), so the sum costs no pass over x.
__device__ __forceinline__ bf16x8 ones8() {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16)1.0f;
    return o;
}
// Fragment `idx` (wave-uniform) of a register array of fragments, by selects: a branch per candidate would cut the
// MFMA stream of the loop into small basic blocks.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NF>
__device__ __forceinline__ bf16x8 pick_frag(const bf16x8 (&f)[NF][2], int idx, int s) {
    // (selected as four dwords: a select on the bf16 vector type is done element by element)
    u32x4 r = __builtin_bit_cast(u32x4, f[0][s]);
#pragma unroll
    for (int i = 1; i < NF; ++i) {
        const u32x4 c = __builtin_bit_cast(u32x4, f[i][s]);
        const bool hit = idx == i;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = hit ? c[j] : r[j];
    }
    return __builtin_bit_cast(bf16x8, r);
}
// One wave's row sums of fragment `a` (rows row0 + 16 a ...): every accumulator row holds the same sums
__device__ __forceinline__ void store_rowsum(float* rowsum, const f32x4& racc, int row0, int a, bool valid, int lane,
                                             int M) {
    const int row = row0 + a * 16 + lane;
    if (valid && lane < 16 && row < M) rowsum[row] = racc[0];
}


// Reduction-major 16-tap gather of a transposed conv's output-grid gradient (the Y operand of that layer's weight
// gradient, never materialised): Y[(b, iy, ix)][(ky * 4 + kx) * C + c] = dout[b][2 iy - 1 + ky][2 ix - 1 + kx][c].
// A lane stages the same two reduction rows (input pixels) of every K tile; their (b, y, x) are decoded once and
// advanced by 64 pixels per K tile with adds and compares.
struct GatherY {
    int b[2], y[2], x[2];
    int u;      // K tile (relative to the slice) the coordinates belong to
};

__device__ __forceinline__ void gather_y_init(GatherY& st, const Gather& g, int pix0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kr = ((i * 8 + wave) * 64 + lane) >> 4;
        const int p = pix0 + kr, hw = g.H * g.W;
        st.b[i] = p / hw;
        const int r = p - st.b[i] * hw;
        st.y[i] = r / g.W;
        st.x[i] = r - st.y[i] * g.W;
    }
    st.u = 0;
}

__device__ __forceinline__ void gather_y_advance(GatherY& st, const Gather& g) {
    const int dy = BK / g.W, dx = BK - dy * g.W;      // 64 pixels further on (host: dy + 1 <= H)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        st.x[i] += dx;
        if (st.x[i] >= g.W) { st.x[i] -= g.W; st.y[i] += 1; }
        st.y[i] += dy;
        if (st.y[i] >= g.H) { st.y[i] -= g.H; st.b[i] += 1; }
    }
    ++st.u;
}

// one 128-column half tile (inside ONE tap: C % 128 == 0) of the gathered operand, image layout of stage_half<true>
__device__ __forceinline__ void stage_half_gather_y(const bf16* dout, const Gather& g, const GatherY& st, int col0,
                                                    char* lds, int wave, int lane) {
    const int t = col0 / g.C, cbase = col0 - t * g.C;   // wave-uniform
    const int ky = t >> 2, kx = t & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int base = (i * 8 + wave) * 64;
        const int L = base + lane;
        const int kr = L >> 4, s16 = L & 15;
        const int unit = (s16 >> 1) ^ kmaj_key(kr);
        const int cc = (unit * 2 + (s16 & 1)) * 8;
        const int oy = 2 * st.y[i] - 1 + ky, ox = 2 * st.x[i] - 1 + kx;
        const bool ok = oy >= 0 && oy < 2 * g.H && ox >= 0 && ox < 2 * g.W;
        const bf16* src = ok ? dout + (((size_t)st.b[i] * 2 * g.H + oy) * 2 * g.W + ox) * g.C + cbase + cc
                             : g.zero + (cc & 56);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + base * 16), 16, 0, 0);
    }
}

template <bool XKMAJ, bool YKMAJ, int EPI, bool GY = false, bool SPLIT3 = false>
__global__ __launch_bounds__(T2_THREADS, 2) void gemm256_kernel(Operand X, Operand Y, int M, int N, int K,
                                                                int tiles_n, int ntiles, int ksplit_tiles, Epi e,
                                                                const vp_group_entry* gtab, int ngtab,
                                                                Gather gth = Gather{}) {
    __shared__ __attribute__((aligned(16))) char lds[2][4][HALF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;

    // 1-D grid over (K slice, tile), slice-major: the contiguous chunk of work ids an XCD receives is a
    // block of neighbouring tiles of ONE K slice, which walk the same operand panels in step (L2 reuse)
    int work = xcd_remap(blockIdx.x, gridDim.x);
    float* rowsum = nullptr;
    if constexpr (XKMAJ && YKMAJ && EPI == VP_EPI_F32) {
        if (gtab) rowsum = select_problem(gtab, ngtab, lane, work, X, Y, M, N, tiles_n, ntiles, e);
    }
    const int slice = work / ntiles;
    const int t = work - slice * ntiles;
    const int tiles_m = ntiles / tiles_n;
    const int gsz = 8 * tiles_n, gid = t / gsz, tin = t - gid * gsz;
    const int gm = min(tiles_m - gid * 8, 8);
    const int i0 = (gid * 8 + tin % gm) * T2, j0 = (tin / gm) * T2;
    const int nk_all = K / BK;
    const int kt0 = slice * ksplit_tiles;
    const int kt1 = min(nk_all, kt0 + ksplit_tiles);
    const int nk = kt1 - kt0;

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr bool ROWSUM = XKMAJ && YKMAJ && EPI == VP_EPI_F32;
    const bool do_rs = ROWSUM && rowsum != nullptr && j0 == 0;   // workgroup-uniform; fragment a goes to wave a % 4
    // row sums: wave wc sums fragment wc of the group's rows 0-63 (phase 0) and of its rows 64-127 (phase 2)
    f32x4 rlo = f32x4{0.f, 0.f, 0.f, 0.f}, rhi = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = ones8();

    // the K loop exists twice (generic lambda): with and without the row-sum MFMAs, so that the tiles
    // without row sums (most of them) run the loop that has no trace of them
    auto k_loop = [&](auto rs_tag) {
        constexpr bool RS = decltype(rs_tag)::value;
        // slot 0/1 = X halves, 2/3 = Y halves; K tile u (relative) lives in stage u & 1
        GatherY gy[2];
        if constexpr (GY) {
            gather_y_init(gy[0], gth, kt0 * BK, wave, lane);
            gy[1] = gy[0];
        }
        auto stage = [&](int slot, int u) {
            const int uu = u < nk ? u : nk - 1;
            const int k0 = (kt0 + uu) * BK;
            char* dst = lds[u & 1][slot];
            if (slot < 2) {
                stage_half<XKMAJ>(X, i0 + slot * 128, k0, dst, wave, lane);
            } else if constexpr (GY) {
                GatherY& st = gy[slot - 2];              // each Y half walks the K tiles in order
                if (st.u < uu) gather_y_advance(st, gth);
                stage_half_gather_y(Y.p, gth, st, j0 + (slot - 2) * 128, dst, wave, lane);
            } else {
                stage_half<YKMAJ>(Y, j0 + (slot - 2) * 128, k0, dst, wave, lane);
            }
        };
        stage(0, 0);
        stage(1, 0);
        stage(2, 0);
        stage(3, 0);
        stage(2, 1);
        VP_WAIT_VM(2);
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one interval behind

        const int ycol = (wc & 1) * 64;
#pragma unroll 1
        for (int u = 0; u < nk; ++u) {
            const char* lx = lds[u & 1][grp];
            const char* ly = lds[u & 1][2 + (wc >> 1)];
            bf16x8 xf[4][2], y0[2][2], y1[2][2];
            // ---- phase 0
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int b = 0; b < 2; ++b) y0[b][s] = read_frag<YKMAJ>(ly, ycol + b * 16, s, lane);
#pragma unroll
                for (int a = 0; a < 4; ++a) xf[a][s] = read_frag<XKMAJ>(lx, a * 16, s, lane);
            }
            stage(3, u + 1);
            __builtin_amdgcn_s_barrier();
            if constexpr (YKMAJ) lds_fence(y0);
            if constexpr (XKMAJ) lds_fence(xf);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < (SPLIT3 ? 3 : 2); ++t)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0[b][SPLIT3 ? t == 2 : t], xf[a][SPLIT3 ? t == 1 : t],
                                                                          acc[a][b], 0, 0, 0);
            if constexpr (RS) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    rlo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pick_frag<4>(xf, wc, s), rlo, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 1
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int b = 0; b < 2; ++b) y1[b][s] = read_frag<YKMAJ>(ly, ycol + 32 + b * 16, s, lane);
            stage(0, u + 1);
            __builtin_amdgcn_s_barrier();
            if constexpr (YKMAJ) lds_fence(y1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < (SPLIT3 ? 3 : 2); ++t)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][2 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1[b][SPLIT3 ? t == 2 : t], xf[a][SPLIT3 ? t == 1 : t],
                                                                          acc[a][2 + b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int a = 0; a < 4; ++a) xf[a][s] = read_frag<XKMAJ>(lx, 64 + a * 16, s, lane);
            stage(1, u + 1);
            __builtin_amdgcn_s_barrier();
            if constexpr (XKMAJ) lds_fence(xf);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < (SPLIT3 ? 3 : 2); ++t)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[4 + a][2 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1[b][SPLIT3 ? t == 2 : t], xf[a][SPLIT3 ? t == 1 : t],
                                                                          acc[4 + a][2 + b], 0, 0, 0);
            if constexpr (RS) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    rhi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pick_frag<4>(xf, wc, s), rhi, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 3
            stage(2, u + 2);
            VP_WAIT_VM(4);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < (SPLIT3 ? 3 : 2); ++t)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[4 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0[b][SPLIT3 ? t == 2 : t], xf[a][SPLIT3 ? t == 1 : t],
                                                                          acc[4 + a][b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            VP_WAIT_VM(2);
            __builtin_amdgcn_s_barrier();
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();   // match group 1's extra barrier
        VP_WAIT_VM(0);                                 // no LDS-DMA may outlive the workgroup
    };
    if (nk > 0) {
        if constexpr (ROWSUM) {
            if (do_rs) k_loop(std::true_type{});
            else k_loop(std::false_type{});
        } else {
            k_loop(std::false_type{});
        }
    }
    if constexpr (epi_is_bf16<EPI>()) {
        if (((N | e.ldo) & 7) == 0) {
            __builtin_amdgcn_s_barrier();              // every wave's loads have landed: LDS is free for staging
            run_epilogue_staged<EPI, 8>(e, acc, i0 + grp * 128, j0 + wc * 64, lane, M, N,
                                        &lds[0][0][0] + wave * STG_BYTES);
            return;
        }
    }
    if constexpr (ROWSUM) {
        if (do_rs) {
            store_rowsum(rowsum, rlo, i0 + grp * 128, wc, true, lane, M);
            store_rowsum(rowsum, rhi, i0 + grp * 128, 4 + wc, true, lane, M);
        }
    }
    if constexpr (EPI == VP_EPI_RESID_F32 && !SPLIT3 && !GY) {
        if (VP_FOLD(e.ln_stats_out)) {
            __builtin_amdgcn_s_barrier();              // every wave's loads have landed: LDS is free
            resid_stats_epilogue<8, 4, 4>(e, acc, nullptr, i0 + grp * 128, j0 + wc * 64, lane, M, N, &lds[0][0][0], wave,
                                          grp, wc, j0 / T2, STG_BYTES);
            return;
        }
    }
    run_epilogue<EPI, 8, 4>(e, acc, i0 + grp * 128, j0 + wc * 64, lane, M, N, slice);
}

// =====================================================================================
// 192 x 192 x 64 tile, 512 threads, one workgroup per CU, three LDS stages
// =====================================================================================
// Same ping-pong schedule as the 256 x 256 kernel for problems whose extents are multiples of 192
// but not of 256 (12288 x 768: 256 tiles = exactly one per CU).  Wave (g, wc) owns the 96 x 48
// block at rows 96 g, columns 48 wc: 6 x 3 fragments, two phases of 18 MFMAs per K tile.
// A K tile is six 64-row staging units (X0 X1 X2 Y0 Y1 Y2, 8 KiB each, one LDS-DMA instruction per
// thread); with three stages every load issued during K tile u targets K tile u + 2, whose stage
// was last read during K tile u - 1 ({X0, X1, Y0} go in phase 0 because group 1 is still reading
// X2 one interval earlier), and ONE counted wait per K tile -- vmcnt(6) at the end of R(u,1) --
// followed by a barrier makes K tile u + 1 visible to both groups.
constexpr int T3 = 192, UNIT_BYTES = 64 * BK * 2, STAGE3_BYTES = 6 * UNIT_BYTES;

__device__ __forceinline__ int kmaj64_key(int kr) { return ((kr >> 1) & 1) | (((kr >> 3) & 1) << 1); }

template <bool KMAJ>
__device__ __forceinline__ void stage_unit(const Operand& op, int r0, int k0, char* lds, int wave, int lane) {
    const int base = wave * 64;
    const int L = base + lane;       // 16-byte slot of the 8 KiB unit
    const bf16* src;
    if constexpr (!KMAJ) {
        const int r = L >> 3, c = (L & 7) ^ (r & 7);
        int row = r0 + r;
        row = row < op.rows ? row : op.rows - 1;
#ifdef VP_ABLATE_SAMETILE
        row %= 192;          // ablation build: every tile reads the operand rows of tile 0 (results wrong; operands L2-resident)
#endif
        src = op.p + (size_t)row * op.ld + k0 + c * 8;
    } else {             // [64 k rows][64 columns], 32-byte units XOR-swizzled for the transposed read
        const int kr = L >> 3, s16 = L & 7;
        const int unit = (s16 >> 1) ^ kmaj64_key(kr);
        int col = r0 + (unit * 2 + (s16 & 1)) * 8;
        col = col <= op.rows - 8 ? col : op.rows - 8;
#ifdef VP_ABLATE_SAMETILE
        col %= 192;
#endif
        src = op.p + (size_t)(k0 + kr) * op.ld + col;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + base * 16), 16, 0, 0);
}

// Per-image operand routing (mixture-of-experts FFN): the Y rows (k-contiguous Y: output columns; k-major Y:
// reduction rows) at or beyond `split` come from the expert of the image the row tile belongs to, `delta`
// elements further on in memory.
struct Route {
    const int* expert;        // [M / rows_per_image] expert index of every image; nullptr = no routing
    const long* delta;        // [experts] element offset of expert e's rows relative to the shared matrix
    const long* bias_delta;   // [experts] the same for the bias vector (or nullptr)
    int split;
};

template <bool KMAJ>
__device__ __forceinline__ void stage_unit_routed(const Operand& op, int r0, int k0, char* lds, int wave, int lane,
                                                  int split, long delta) {
    const int base = wave * 64;
    const int L = base + lane;
    const bf16* src;
    if constexpr (!KMAJ) {
        const int r = L >> 3, c = (L & 7) ^ (r & 7);
        int row = r0 + r;
        row = row < op.rows ? row : op.rows - 1;
        src = op.p + (size_t)row * op.ld + k0 + c * 8 + (row >= split ? delta : 0);
    } else {
        const int kr = L >> 3, s16 = L & 7;
        const int unit = (s16 >> 1) ^ kmaj64_key(kr);
        int col = r0 + (unit * 2 + (s16 & 1)) * 8;
        col = col <= op.rows - 8 ? col : op.rows - 8;
        const int kk = k0 + kr;
        src = op.p + (size_t)kk * op.ld + col + (kk >= split ? delta : 0);
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + base * 16), 16, 0, 0);
}

template <int N_>
using ic = std::integral_constant<int, N_>;

template <int N_>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// fragment `f` (16 rows / columns) of a 192-wide operand tile made of three 64-wide units
template <bool KMAJ>
__device__ __forceinline__ bf16x8 read_frag3(const char* op_units, int f, int s, int lane) {
    const int rbase = (f & 3) * 16;
    if constexpr (!KMAJ) {
        return read_frag<false>(op_units + (f >> 2) * UNIT_BYTES, rbase, s, lane);
    } else {
        // (as in read_frag<true>: one per-lane address per 16-column unit, everything else in the offset field)
        const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, u16 = rbase >> 4, kq = 8 * g + q;
        const char* base = op_units + (f >> 2) * UNIT_BYTES + kq * 128 + ((u16 ^ kmaj64_key(kq)) << 5) + (p << 3);
        bf16x4 lo = lds_read_tr_raw(base, s * 4096);
        bf16x4 hi = lds_read_tr_raw(base, s * 4096 + 512);
        return cat8(lo, hi);
    }
}

// ROUTE: per-image Y routing (see Route; a row tile is exactly one image of 192 tokens).
// KMAP (grouped weight gradients only): a problem may reduce over a SUBSET of the images whose rows make up the
// K extent: bit b of its image_mask words selects image b (tiles_per_image K tiles each; no bit set = the output
// is written as zeros).  The K loop walks the set bits with scalar ALU work only (no memory access in the loop).
// NU = 64-column units of the Y tile: 3 = the 192 x 192 tile, 2 = a 192 x 128 tile (a wave owns 96 x 32: 6 x 2 fragments)
// for products whose 192-wide column tiles would not fill the chip (N = 1024 at 6144 rows: 256 tiles of 192 x 128).
template <bool XKMAJ, bool YKMAJ, int EPI, bool ROUTE = false, bool KMAP = false, bool SPLIT3 = false, int NU = 3>
__global__ __launch_bounds__(T2_THREADS, 2) void gemm192_kernel(Operand X, Operand Y, int M, int N, int K,
                                                                int tiles_n, int ntiles, int ksplit_tiles, Epi e,
                                                                const vp_group_entry* gtab, int ngtab,
                                                                Route rt = Route{nullptr, nullptr, nullptr, 0}) {
    static_assert(NU == 3 || (NU == 2 && !ROUTE && !KMAP), "routed / masked launches use the 192-wide tile");
    constexpr int STAGE_BYTES = (3 + NU) * UNIT_BYTES, WCOLS = 16 * NU, TN = 64 * NU;
    __shared__ __attribute__((aligned(16))) char lds[3][STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;

    // 1-D grid over (K slice, tile), slice-major: the contiguous chunk of work ids an XCD receives is a
    // block of neighbouring tiles of ONE K slice, which walk the same operand panels in step (L2 reuse)
    // (masked launches mix full-length and short problems: dealing the tile list out to the XCDs round-robin
    // gives every XCD the same mix, where contiguous eighths would not)
    int work = KMAP ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    float* rowsum = nullptr;
    int pidx = 0;
    if constexpr (XKMAJ && YKMAJ && EPI == VP_EPI_F32) {
        if (gtab) rowsum = select_problem(gtab, ngtab, lane, work, X, Y, M, N, tiles_n, ntiles, e, &pidx);
    }
    const int slice = work / ntiles;
    const int t = work - slice * ntiles;
    const int tiles_m = ntiles / tiles_n;
    const int gsz = 8 * tiles_n, gid = t / gsz, tin = t - gid * gsz;
    const int gm = min(tiles_m - gid * 8, 8);
    const int i0 = (gid * 8 + tin % gm) * T3, j0 = (tin / gm) * TN;
    const int nk_all = K / BK;
    const int kt0 = slice * ksplit_tiles;
    const int kt1 = min(nk_all, kt0 + ksplit_tiles);
    int nk = kt1 - kt0;
    VP_CLK_BEGIN();
    // image iterator of a masked problem: `mcur` = unvisited bits of mask word `mword`, (mimg, msub) = position
    const unsigned long long* mwords = nullptr;
    unsigned long long mcur = 0;
    int mword = 0, mnwords = 0, mimg = 0, msub = 0, mtpi = 1, mk0 = 0;
    if constexpr (KMAP) {
        if (gtab && gtab[pidx].image_mask) {
            mwords = gtab[pidx].image_mask;
            mtpi = gtab[pidx].tiles_per_image;
            mnwords = (nk_all / mtpi + 63) / 64;
            int bits = 0;
            for (int w = 0; w < mnwords; ++w) bits += __popcll(mwords[w]);
            nk = min(bits * mtpi, nk_all);
            mcur = mwords[0];
            msub = mtpi;       // "current image used up": the first call fetches the first set bit
        }
    }
    // first row of the next K tile in staging order (tiles are staged strictly in order 0, 1, 2, ...; past the
    // end the last tile is staged again, as the unmasked clamp does)
    auto masked_next_k0 = [&]() {
        if (msub == mtpi) {
            while (mcur == 0 && mword + 1 < mnwords) mcur = mwords[++mword];
            if (mcur != 0) {
                mimg = mword * 64 + __builtin_ctzll(mcur);
                mcur &= mcur - 1;
                msub = 0;
            } else {
                return mk0;
            }
        }
        mk0 = (mimg * mtpi + msub) * BK;
        ++msub;
        return mk0;
    };
    long ydelta = 0;
    if constexpr (ROUTE) {
        const int ex = rt.expert[i0 / T3];
        ydelta = rt.delta[ex];
        if (e.bias && rt.bias_delta && j0 + wc * WCOLS >= rt.split) e.bias += rt.bias_delta[ex];
    }

    f32x4 acc[6][NU];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < NU; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // residual / position addend of this wave's block: in flight under the whole K loop
    f32x4 pre[EPI == VP_EPI_RESID_F32 ? 6 : 1][NU];
    if constexpr (EPI == VP_EPI_RESID_F32) load_addend<6, NU>(e, pre, i0 + grp * 96, j0 + wc * WCOLS, lane, M, N);
    constexpr bool ROWSUM = XKMAJ && YKMAJ && EPI == VP_EPI_F32;
    const bool do_rs = ROWSUM && rowsum != nullptr && j0 == 0;   // workgroup-uniform; fragment a goes to wave a % 4
    // row sums: fragment a of the group's six goes to wave a % 4 -- phase 0 (fragments 0-2): wave wc sums fragment wc
    // (wave 3 none); phase 1 (fragments 3-5): wave wc sums fragment 3 + (wc + 1) % 4 (wave 2 none).  A wave without a
    // fragment multiplies by zeros instead of ones: no branch in the MFMA stream.
    f32x4 r0 = f32x4{0.f, 0.f, 0.f, 0.f}, r1 = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rs_a1 = (wc + 1) & 3;
    bf16x8 ones0 = ones8(), ones1 = ones8();
    if (wc == 3) ones0 = bf16x8{};
    if (rs_a1 == 3) ones1 = bf16x8{};

    // the K loop exists twice (generic lambda): with and without the row-sum MFMAs, so that the tiles
    // without row sums (most of them) run the loop that has no trace of them
    auto k_loop = [&](auto rs_tag) {
        constexpr bool RS = decltype(rs_tag)::value;
        // unit 0-2 = X rows 0-63 / 64-127 / 128-191, 3-5 = Y columns likewise; K tile u lives in stage u % 3
        int kstage = 0;   // first reduction row of the K tile being staged
        auto pick = [&](int u) {
            kstage = (kt0 + (u < nk ? u : nk - 1)) * BK;
            if constexpr (KMAP) {
                if (mwords) kstage = masked_next_k0();
            }
        };
        auto stage = [&](int unit, int u, int st) {
#ifdef VP_ABLATE_NOLOAD
            if (u >= 2) return;     // ablation build: only the pipeline fill is loaded (timing of a loop without operand traffic)
#endif
            const int k0 = kstage;
            char* dst = lds[st] + unit * UNIT_BYTES;
            if (unit < 3) stage_unit<XKMAJ>(X, i0 + unit * 64, k0, dst, wave, lane);
            else if constexpr (ROUTE) stage_unit_routed<YKMAJ>(Y, j0 + (unit - 3) * 64, k0, dst, wave, lane, rt.split, ydelta);
            else stage_unit<YKMAJ>(Y, j0 + (unit - 3) * 64, k0, dst, wave, lane);
        };
        pick(0);
#pragma unroll
        for (int un = 0; un < 3 + NU; ++un) stage(un, 0, 0);
        pick(1);
#pragma unroll
        for (int un = 0; un < 3 + NU; ++un) stage(un, 1, 1);
        wait_vm<3 + NU>();
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one interval behind

        int cur = 0, nxt2 = 2;                        // stage of K tile u, of K tile u + 2
#pragma unroll 1
        for (int u = 0; u < nk; ++u) {
            pick(u + 2);
            const char* lx = lds[cur];
            const char* ly = lds[cur] + 3 * UNIT_BYTES;
            bf16x8 xf[3][2], yf[NU][2];
            // ---- phase 0: rows 0-47 of the wave's block
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int b = 0; b < NU; ++b) yf[b][s] = read_frag3<YKMAJ>(ly, NU * wc + b, s, lane);
#pragma unroll
                for (int a = 0; a < 3; ++a) xf[a][s] = read_frag3<XKMAJ>(lx, 6 * grp + a, s, lane);
            }
            stage(0, u + 2, nxt2);
            stage(1, u + 2, nxt2);
            stage(3, u + 2, nxt2);
            __builtin_amdgcn_s_barrier();
            if constexpr (YKMAJ) lds_fence(yf);
            if constexpr (XKMAJ) lds_fence(xf);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < (SPLIT3 ? 3 : 2); ++t)
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < NU; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[b][SPLIT3 ? t == 2 : t], xf[a][SPLIT3 ? t == 1 : t],
                                                                          acc[a][b], 0, 0, 0);
            if constexpr (RS) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    r0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones0, pick_frag<3>(xf, wc, s), r0, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 1: rows 48-95
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int a = 0; a < 3; ++a) xf[a][s] = read_frag3<XKMAJ>(lx, 6 * grp + 3 + a, s, lane);
            stage(2, u + 2, nxt2);
            stage(4, u + 2, nxt2);
            if constexpr (NU == 3) stage(5, u + 2, nxt2);
            wait_vm<3 + NU>();
            __builtin_amdgcn_s_barrier();
            if constexpr (XKMAJ) lds_fence(xf);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < (SPLIT3 ? 3 : 2); ++t)
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < NU; ++b)
                        acc[3 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[b][SPLIT3 ? t == 2 : t], xf[a][SPLIT3 ? t == 1 : t],
                                                                          acc[3 + a][b], 0, 0, 0);
            if constexpr (RS) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    r1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones1, pick_frag<3>(xf, rs_a1, s), r1, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            cur = cur == 2 ? 0 : cur + 1;
            nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();   // match group 1's extra barrier
        VP_WAIT_VM(0);                                 // no LDS-DMA may outlive the workgroup
    };
    if (nk > 0) {
        if constexpr (ROWSUM) {
            if (do_rs) k_loop(std::true_type{});
            else k_loop(std::false_type{});
        } else {
            k_loop(std::false_type{});
        }
    }
    VP_CLK_END("gemm192");
    if constexpr (epi_is_bf16<EPI>()) {
        if (((N | e.ldo) & 7) == 0) {
            __builtin_amdgcn_s_barrier();              // every wave's loads have landed: LDS is free for staging
            run_epilogue_staged<EPI, 6, NU, 3>(e, acc, i0 + grp * 96, j0 + wc * WCOLS, lane, M, N,
                                               &lds[0][0] + wave * (48 * (NU * 32 + 16)));
            VP_CLK_FINAL("gemm192", work);
            return;
        }
    }
    if constexpr (ROWSUM) {
        if (do_rs) {
            store_rowsum(rowsum, r0, i0 + grp * 96, wc, wc < 3, lane, M);
            store_rowsum(rowsum, r1, i0 + grp * 96, 3 + rs_a1, rs_a1 < 3, lane, M);
        }
    }
    if constexpr (EPI == VP_EPI_RESID_F32) {
        if constexpr (!ROUTE && !KMAP && !SPLIT3) {
            if (VP_FOLD(e.ln_stats_out)) {
                __builtin_amdgcn_s_barrier();              // every wave's loads have landed: LDS is free
                resid_stats_epilogue<6, NU, 3>(e, acc, pre, i0 + grp * 96, j0 + wc * WCOLS, lane, M, N, &lds[0][0], wave,
                                               grp, wc, j0 / TN, 48 * (NU * 32 + 16));
                VP_CLK_FINAL("gemm192", work);
                return;
            }
        }
        run_epilogue<EPI, 6, NU>(e, acc, i0 + grp * 96, j0 + wc * WCOLS, lane, M, N, slice, pre);
    } else {
        run_epilogue<EPI, 6, NU>(e, acc, i0 + grp * 96, j0 + wc * WCOLS, lane, M, N, slice);
    }
    VP_CLK_FINAL("gemm192", work);
}


// =====================================================================================
// 192 x 192 x 64 tile with the K tile split between the two wave groups (tile_mode 7)
// =====================================================================================
// Wave (h, r, c) owns the 96 x 96 block (r, c) of the tile for the 32-wide sub-step h of every K tile: 6 + 6 fragment
// reads feed 36 MFMAs (the 96 x 48 waves of gemm192_kernel read 12 + 6 for 36), one interval is 36 MFMAs long instead
// of 18 and a K tile costs two barriers instead of four.  Measured on the bare structure (tools/probes/mfma_lds_probe):
// 2050 TFLOP/s against 1580 for the 96 x 48 layout, i.e. the rate of an MFMA loop without any LDS traffic.  The price is
// paid once per output tile: the partial sums of the two sub-steps meet through LDS (each wave hands 48 of its 96 rows
// to its partner and finishes the other 48: 144 KiB, the three stages), so the layout is for long reductions -- the
// grouped weight gradients and the K >= 2304 data-gradient / forward products.
// Group h = 1 runs one barrier interval behind group h = 0 (its reads and LDS-DMA issue under the other group's MFMAs).
// Per K tile u and wave: R(u) = [12 fragment reads of K tile u, LDS-DMA of K tile u + 2], barrier, C(u) = [36 MFMAs], barrier.
// K tile u + 1 has to be complete before the barrier after which group 0 reads it: group 0 waits for its own loads at the
// end of C(u), group 1 -- which reaches that same barrier at the end of its R(u) -- there.  The fragment reads are waited
// for BEFORE the barrier that ends R: the stage they come from is overwritten by loads issued in the next interval.
template <bool XKMAJ, bool YKMAJ, int EPI>
__global__ __launch_bounds__(T2_THREADS, 2) void gemm192k_kernel(Operand X, Operand Y, int M, int N, int K, int tiles_n,
                                                                 int ntiles, Epi e, const vp_group_entry* gtab, int ngtab) {
    __shared__ __attribute__((aligned(16))) char lds[3][STAGE3_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 2, q = wave & 3, r = q >> 1, c = q & 1;
    int work = xcd_remap(blockIdx.x, gridDim.x);
    float* rowsum = nullptr;
    if constexpr (XKMAJ && YKMAJ && EPI == VP_EPI_F32) {
        if (gtab) rowsum = select_problem(gtab, ngtab, lane, work, X, Y, M, N, tiles_n, ntiles, e);
    }
    const int t = work;
    const int tiles_m = ntiles / tiles_n;
    const int gsz = 8 * tiles_n, gid = t / gsz, tin = t - gid * gsz;
    const int gm = min(tiles_m - gid * 8, 8);
    const int i0 = (gid * 8 + tin % gm) * T3, j0 = (tin / gm) * T3;
    const int nk = K / BK;
    VP_CLK_BEGIN();

    f32x4 acc[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr bool ROWSUM = XKMAJ && YKMAJ && EPI == VP_EPI_F32;
    const bool do_rs = ROWSUM && rowsum != nullptr && j0 == 0;
    // row sums (bias gradients): one f32 per lane and X fragment, += the lane's 8 reduction elements of its row by four
    // v_dot2c_f32_bf16 against (1, 1) per K tile -- VALU work in the shadow of the MFMAs, six registers (an MFMA against
    // a ones operand, as in the kernels above, costs twelve accumulator registers more than this kernel has)
    float rsv[ROWSUM ? 6 : 1];
#pragma unroll
    for (int a = 0; a < (ROWSUM ? 6 : 1); ++a) rsv[a] = 0.f;

    auto k_loop = [&](auto rs_tag) {
        constexpr bool RS = decltype(rs_tag)::value;
        auto stage_all = [&](int u, int st) {
            const int k0 = (u < nk ? u : nk - 1) * BK;
#pragma unroll
            for (int un = 0; un < 6; ++un) {
                char* dst = lds[st] + un * UNIT_BYTES;
                if (un < 3) stage_unit<XKMAJ>(X, i0 + un * 64, k0, dst, wave, lane);
                else stage_unit<YKMAJ>(Y, j0 + (un - 3) * 64, k0, dst, wave, lane);
            }
        };
        stage_all(0, 0);
        stage_all(1, 1);
        wait_vm<6>();
        __builtin_amdgcn_s_barrier();
        if (h == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one interval behind
        int cur = 0, nxt2 = 2;
#pragma unroll 1
        for (int u = 0; u < nk; ++u) {
            const char* lx = lds[cur];
            const char* ly = lds[cur] + 3 * UNIT_BYTES;
            bf16x8 xf[6], yf[6];
            // (a k-major image keeps sub-step s of a unit 4096 bytes further on: the transposed read takes its offset as
            // an immediate, so the wave's sub-step goes into the base address)
#pragma unroll
            for (int b = 0; b < 6; ++b) yf[b] = read_frag3<YKMAJ>(ly + (YKMAJ ? h * 4096 : 0), 6 * c + b, YKMAJ ? 0 : h, lane);
#pragma unroll
            for (int a = 0; a < 6; ++a)
                xf[a] = read_frag3<XKMAJ>(lx + (XKMAJ ? h * 4096 : 0), 6 * r + (a + 3 * h) % 6, XKMAJ ? 0 : h, lane);
            stage_all(u + 2, nxt2);
            lds_fence(yf);
            lds_fence(xf);
            if (h == 1) wait_vm<6>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[b], xf[a], acc[a][b], 0, 0, 0);
            if constexpr (RS) {
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const u32x4 w = __builtin_bit_cast(u32x4, xf[a]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        asm("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(rsv[a]) : "v"(w[i]), "s"(0x3f803f80u));
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (h == 0) wait_vm<6>();
            __builtin_amdgcn_s_barrier();
            cur = cur == 2 ? 0 : cur + 1;
            nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
        }
        if (h == 0) __builtin_amdgcn_s_barrier();   // match group 1's extra barrier
        VP_WAIT_VM(0);                               // no LDS-DMA may outlive the K loop: the stages are reused below
    };
    if (nk > 0) {
        if constexpr (ROWSUM) {
            if (do_rs) k_loop(std::true_type{});
            else k_loop(std::false_type{});
        } else {
            k_loop(std::false_type{});
        }
    }
    // ---- the two sub-step halves meet: wave (0, q) keeps rows 0-47 of its block, wave (1, q) rows 48-95; each hands the
    // other half (18 fragments, 1 KiB each, lane-linear) to its partner through LDS
    VP_CLK_END("gemm192k");
    __builtin_amdgcn_s_barrier();                    // every wave's loads have landed and every read of the stages is done
    char* flat = &lds[0][0];
    f32x4* mine = reinterpret_cast<f32x4*>(flat + wave * 18432) + lane;
    const f32x4* theirs = reinterpret_cast<const f32x4*>(flat + (wave ^ 4) * 18432) + lane;
    // (a wave of group 1 keeps its row fragments rotated by three -- xf[a] is fragment (a + 3) % 6 of the block -- so that
    // for both groups acc[0..2] is the half that stays and acc[3..5] the half that goes: no register index depends on h)
    const int row0 = i0 + r * 96 + h * 48, col0 = j0 + c * 96;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) mine[(a * 6 + b) * 64] = acc[3 + a][b];
    // the residual / position addend of the half that stays: requested now, into the registers the other half has just
    // left, so that it arrives under the exchange
    f32x4 pre[EPI == VP_EPI_RESID_F32 ? 3 : 1][6];
    if constexpr (EPI == VP_EPI_RESID_F32) load_addend<3, 6>(e, pre, row0, col0, lane, M, N);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[a][b] += theirs[(a * 6 + b) * 64];
    f32x4 (&fin)[3][6] = reinterpret_cast<f32x4 (&)[3][6]>(acc);
    if constexpr (ROWSUM) {
        if (do_rs) {                                 // (workgroup-uniform)
            __builtin_amdgcn_s_barrier();            // the exchange has been read
            // lane (g, li) holds the sum over its k chunks of row li: add the four lane groups, then the two sub-steps
            // (the waves with c = 1 hold the same sums as their c = 0 neighbours and stay out)
            float* part = reinterpret_cast<float*>(flat);
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                float v = rsv[a];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (c == 0 && lane < 16) part[((h * 2 + r) * 6 + (a + 3 * h) % 6) * 16 + lane] = v;   // by block fragment
            }
            __builtin_amdgcn_s_barrier();
            if (h == 0 && c == 0 && lane < 16) {
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const int row = i0 + r * 96 + a * 16 + lane;
                    if (row < M) rowsum[row] = part[(r * 6 + a) * 16 + lane] + part[((2 + r) * 6 + a) * 16 + lane];
                }
            }
        }
    }
    if constexpr (epi_is_bf16<EPI>()) {
        if (((N | e.ldo) & 7) == 0) {
            __builtin_amdgcn_s_barrier();            // the exchange has been read: LDS is free for the output transpose
            run_epilogue_staged<EPI, 3, 6, 3>(e, fin, row0, col0, lane, M, N, flat + wave * (48 * 208));
            return;
        }
    }
    if constexpr (EPI == VP_EPI_RESID_F32) run_epilogue<EPI, 3, 6>(e, fin, row0, col0, lane, M, N, 0, pre);
    else run_epilogue<EPI, 3, 6>(e, fin, row0, col0, lane, M, N);
    VP_CLK_FINAL("gemm192k", work);
}


// =====================================================================================
// Persistent 192 x 192 x 64 kernel: the epilogue of tile i runs under the K loop of tile i + 1
// =====================================================================================
// One workgroup per CU walks a list of output tiles (tile = slot, slot + grid, ...) through ONE continuous K-tile
// pipeline: the LDS-DMA loads issued during the last two K tiles of an output tile already fetch the first two K tiles
// of the next one, so only the first tile of a workgroup pays the pipeline fill.  When an output tile is complete its
// accumulators move to a second register set (`pacc`) and the next K loop starts at once; the pending tile is written
// fragment by fragment (16 x 16 per wave: the epilogue arithmetic + one 8-byte store per lane) in the first 18 phases of
// K tiles 1-9 of that K loop, in the half of each phase where this wave reads LDS and the partner wave of its SIMD
// streams MFMAs: the epilogue's VALU work issues in the shadow of those MFMAs and its HBM traffic shares the memory
// system with operand loads instead of arriving from every CU at once.  The last tile of a workgroup is written the
// ordinary way.
// Counted waits: loads return in order, so "all LDS-DMA loads of the previous window have landed" = at most (6 + the
// global LOADS issued since) operations outstanding; the stores in between only make that wait stricter.  The bias /
// pre-activation loads of the pending tile are issued at the tile switch or six fragments ahead, and always have a
// counted wait of the K loop between them and their use.
// Needs K >= 10 * 64 (one fragment per phase, 18 fragments, after one K tile without).  bf16 outputs (VP_EPI_BF16 / GELU / DGELU), no row remap,
// no column scale.
// Global loads the compiler does not track (it would otherwise wait for EVERY outstanding vector-memory operation,
// the LDS-DMA prefetches included, in front of their first use inside the K loop): the caller guarantees that a
// counted wait lies between the load and loaded(), which ties the register to that point of the program.
__device__ __forceinline__ void gload_raw(bf16x4& dst, const void* p) {
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void gload_raw(f32x4& dst, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <typename T>
__device__ __forceinline__ void loaded(T& v) {
    asm volatile("" : "+v"(v));
}

template <bool XKMAJ, bool YKMAJ, int EPI>
__global__ __launch_bounds__(T2_THREADS, 2) void gemm192p_kernel(Operand X, Operand Y, int M, int N, int K,
                                                                 int tiles_n, int ntiles, Epi e) {
    static_assert(EPI == VP_EPI_BF16 || EPI == VP_EPI_GELU || EPI == VP_EPI_DGELU, "bf16 outputs only");
    __shared__ __attribute__((aligned(16))) char lds[3][STAGE3_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, li = lane & 15;
    const int G = gridDim.x;
    const int tiles_m = ntiles / tiles_n;
    const int nk = K / BK;
    auto coords = [&](int t, int& i0, int& j0) {
        const int gsz = 8 * tiles_n, gid = t / gsz, tin = t - gid * gsz;
        const int gm = min(tiles_m - gid * 8, 8);
        i0 = (gid * 8 + tin % gm) * T3;
        j0 = (tin / gm) * T3;
    };
    int t = xcd_remap(blockIdx.x, G);          // G <= ntiles: every workgroup owns at least one tile
    int i0, j0;
    coords(t, i0, j0);

    f32x4 acc[6][3], pacc[6][3];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = pacc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool pending = false;
    int pi0 = 0, pj0 = 0;
    // per pending tile: this lane's first output element, bias of its 12 columns, pre-activation prefetch ring
    bf16* po = nullptr;
    bf16* po2 = nullptr;
    const bf16* paux = nullptr;
    f32x4 pbias[3];
    bf16x4 uq[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) uq[i] = bf16x4{};
#pragma unroll
    for (int b = 0; b < 3; ++b) pbias[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool two_outputs = EPI == VP_EPI_GELU && e.out2 != nullptr;

    auto stage_at = [&](int unit, int si0, int sj0, int sk, int st) {
        char* dst = lds[st] + unit * UNIT_BYTES;
        if (unit < 3) stage_unit<XKMAJ>(X, si0 + unit * 64, sk, dst, wave, lane);
        else stage_unit<YKMAJ>(Y, sj0 + (unit - 3) * 64, sk, dst, wave, lane);
    };
    auto load_u = [&](bf16x4& dst, int f) {       // pre-activations of fragment f = (a, b) of the pending tile
        const int a = f / 3, b = f - a * 3;
        gload_raw(dst, paux + (size_t)(a * 16) * e.ldaux + b * 16);
    };
    // Epilogue of fragment F of the pending tile (F < 0: nothing).  Only tiles that lie wholly inside the output are
    // deferred, so every store below is unconditional: the number of vector-memory operations a slot issues is a
    // compile-time constant (the counted waits of the K loop rely on it).
    auto slot = [&](auto ftag) {
        constexpr int F = decltype(ftag)::value;
        if constexpr (F >= 0) {
            constexpr int a = F / 3, b = F % 3;
            f32x4 v = pacc[a][b];
            bf16* o = po + (size_t)(a * 16) * e.ldo + b * 16;
            if constexpr (EPI == VP_EPI_DGELU) {
                loaded(uq[F % 6]);
                const bf16x4 u = uq[F % 6];
                if constexpr (F + 6 < 18) load_u(uq[F % 6], F + 6);
                v *= gelu_erf_grad(__builtin_convertvector(u, f32x4));
            } else {
                if constexpr (a == 0) loaded(pbias[b]);
                v = v + pbias[b];
                if constexpr (EPI == VP_EPI_GELU) {
                    if (two_outputs)
                        *reinterpret_cast<bf16x4*>(po2 + (size_t)(a * 16) * e.ldo + b * 16) = pack4(v[0], v[1], v[2], v[3]);
                    v = gelu_erf(v);
                }
            }
            *reinterpret_cast<bf16x4*>(o) = pack4(v[0], v[1], v[2], v[3]);
        }
    };

    // pipeline fill: K tiles 0 and 1 of the first output tile
#pragma unroll
    for (int un = 0; un < 6; ++un) stage_at(un, i0, j0, 0, 0);
#pragma unroll
    for (int un = 0; un < 6; ++un) stage_at(un, i0, j0, BK, 1);
    VP_WAIT_VM(6);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one interval behind

    int cur = 0, nxt2 = 2;                        // stage of K tile u, of K tile u + 2
    while (true) {
        const int tn = t + G;
        const bool has_next = tn < ntiles;
        int ni0 = i0, nj0 = j0;
        if (has_next) coords(tn, ni0, nj0);
        int u = 0;
        // One K tile; S0 / S1 = pending-tile fragments written in its two phases.  Vector-memory operations between
        // two counted waits, in issue order: 3 LDS-DMA, slot S0, slot S1, 3 LDS-DMA -- everything older (the LDS-DMA
        // loads of K tile u + 1 among it) has landed once at most that many operations are outstanding (loads and
        // stores of a wave complete in issue order on this target).
        auto iter = [&](auto s0tag, auto s1tag) {
            constexpr int S0 = decltype(s0tag)::value, S1 = decltype(s1tag)::value;
            constexpr int NSLOT = (S0 >= 0) + (S1 >= 0);
            constexpr int NLOADS = EPI == VP_EPI_DGELU ? (S0 >= 0 && S0 + 6 < 18) + (S1 >= 0 && S1 + 6 < 18) : 0;
            int si0 = i0, sj0 = j0, sk = (u + 2) * BK;        // K tile u + 2, possibly of the next output tile
            if (u + 2 >= nk) {
                if (has_next) { si0 = ni0; sj0 = nj0; sk = (u + 2 - nk) * BK; }
                else sk = (nk - 1) * BK;
            }
            const char* lx = lds[cur];
            const char* ly = lds[cur] + 3 * UNIT_BYTES;
            bf16x8 xf[3][2], yf[3][2];
            // ---- phase 0: rows 0-47 of the wave's block
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int b = 0; b < 3; ++b) yf[b][s] = read_frag3<YKMAJ>(ly, 3 * wc + b, s, lane);
#pragma unroll
                for (int a = 0; a < 3; ++a) xf[a][s] = read_frag3<XKMAJ>(lx, 6 * grp + a, s, lane);
            }
            stage_at(0, si0, sj0, sk, nxt2);
            stage_at(1, si0, sj0, sk, nxt2);
            stage_at(3, si0, sj0, sk, nxt2);
            slot(s0tag);
            __builtin_amdgcn_s_barrier();
            if constexpr (YKMAJ) lds_fence(yf);
            if constexpr (XKMAJ) lds_fence(xf);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[b][s], xf[a][s], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 1: rows 48-95
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int a = 0; a < 3; ++a) xf[a][s] = read_frag3<XKMAJ>(lx, 6 * grp + 3 + a, s, lane);
            slot(s1tag);
            stage_at(2, si0, sj0, sk, nxt2);
            stage_at(4, si0, sj0, sk, nxt2);
            stage_at(5, si0, sj0, sk, nxt2);
            if (EPI == VP_EPI_GELU && two_outputs) wait_vm<6 + 2 * NSLOT>();
            else wait_vm<6 + NSLOT + NLOADS>();
            __builtin_amdgcn_s_barrier();
            if constexpr (XKMAJ) lds_fence(xf);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        acc[3 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[b][s], xf[a][s], acc[3 + a][b], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            cur = cur == 2 ? 0 : cur + 1;
            nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
            ++u;
        };
        if (pending) {
            iter(ic<-1>{}, ic<-1>{});      // the loads issued at the tile switch land behind this K tile's wait
            iter(ic<0>{}, ic<1>{});
            iter(ic<2>{}, ic<3>{});
            iter(ic<4>{}, ic<5>{});
            iter(ic<6>{}, ic<7>{});
            iter(ic<8>{}, ic<9>{});
            iter(ic<10>{}, ic<11>{});
            iter(ic<12>{}, ic<13>{});
            iter(ic<14>{}, ic<15>{});
            iter(ic<16>{}, ic<17>{});
        }
#pragma unroll 1
        while (u < nk) iter(ic<-1>{}, ic<-1>{});
        // The tile is complete.  A tile that lies wholly inside the output becomes the pending one; a ragged last
        // column tile is written at once with the guarded epilogue (registers -> global, no LDS: the stages hold the
        // next tile's prefetch).
        pending = j0 + T3 <= N;
        if (pending) {
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) pacc[a][b] = acc[a][b];
            pi0 = i0;
            pj0 = j0;
        } else {
            run_epilogue<EPI, 6, 3>(e, acc, i0 + grp * 96, j0 + wc * 48, lane, M, N);
        }
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!has_next) break;
        if (pending) {
            // M is a multiple of the tile height (checked by the launcher): every row exists
            const int row = pi0 + grp * 96 + li, col = pj0 + wc * 48 + g * 4;
            po = (bf16*)e.out + (size_t)row * e.ldo + col;
            if constexpr (EPI == VP_EPI_GELU) po2 = two_outputs ? (bf16*)e.out2 + (size_t)row * e.ldo + col : nullptr;
            if constexpr (EPI == VP_EPI_DGELU) {
                paux = (const bf16*)e.aux + (size_t)row * e.ldaux + col;
#pragma unroll
                for (int f = 0; f < 6; ++f) load_u(uq[f], f);
            } else if (e.bias) {
#pragma unroll
                for (int b = 0; b < 3; ++b) gload_raw(pbias[b], e.bias + col + b * 16);
            }
        }
        t = tn;
        i0 = ni0;
        j0 = nj0;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // match group 1's extra barrier
    VP_WAIT_VM(0);                                 // no LDS-DMA may outlive the workgroup
    if (pending) {
        if (((N | e.ldo) & 7) == 0) {
            __builtin_amdgcn_s_barrier();              // every wave's loads have landed: LDS is free for staging
            run_epilogue_staged<EPI, 6, 3, 3>(e, pacc, pi0 + grp * 96, pj0 + wc * 48, lane, M, N,
                                              &lds[0][0] + wave * (48 * 112));
        } else {
            run_epilogue<EPI, 6, 3>(e, pacc, pi0 + grp * 96, pj0 + wc * 48, lane, M, N);
        }
    }
}

inline int ln_slots_of(int N, int tile_mode) {
    const int w = tile_mode == 2 ? T2 : tile_mode == 3 ? T3 : tile_mode == 6 ? 128 : 0;
    return w ? (N + w - 1) / w : 0;
}

inline int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}

// Heuristic: the big tile pays when its workgroups keep most CUs busy in whole rounds.
inline bool use_big_tiles(int M, int N, int K, int units, int cus) {
    if (M < 2 * T2 || N < T2 || K < 4 * BK) return false;
    const int rounds = (units + cus - 1) / cus;
    return (double)units / (double)(rounds * cus) >= 0.70;
}

template <bool XK, bool YK>
int launch_layout(const vp_gemm_args* a, hipStream_t st) {
    Operand X{(const bf16*)a->x, a->ldx, a->M}, Y{(const bf16*)a->y, a->ldy, a->N};
    Epi e{a->epilogue, a->out, a->ldo, a->out2, a->bias, a->resid, a->rowscale, a->rows_per_scale,
          a->aux, a->ldaux, a->pos, a->pos_rows, a->colscale,
          a->out_row_div, a->out_row_hi, a->out_row_lo, a->out_row_off, a->split_stride};
    e.ln_stats = a->ln_stats;
    e.ln_stats_out = a->ln_stats_out;
    e.ln_slots = a->ln_slots;
    e.ln_inv_n = 1.0f / (float)a->K;
    e.ln_eps = a->ln_eps;
    if (a->ln_stats) {
        // consumer of a folded LayerNorm: the reduction runs over the normalised channels
        if ((a->epilogue != VP_EPI_BF16 && a->epilogue != VP_EPI_GELU) || !a->colscale || !a->bias || a->ln_slots < 1 ||
            a->tile_mode == 5 || a->out_row_div > 0 || XK)
            return VP_ERR_UNSUPPORTED;
    }
    if (a->ln_stats_out) {
        // producer: the kernels whose column waves share their rows, on an explicit tile schedule (it fixes ln_slots)
        if (a->epilogue != VP_EPI_RESID_F32 || !a->out2 || XK || a->out_row_div > 0 || ((a->N | a->ldo) & 7) != 0 ||
            (a->tile_mode != 2 && a->tile_mode != 3 && a->tile_mode != 6) ||
            a->ln_slots != ln_slots_of(a->N, a->tile_mode))
            return VP_ERR_UNSUPPORTED;
    }
    const int tiles_m = (a->M + BM - 1) / BM, tiles_n = (a->N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    const int nk = a->K / BK;
    int split = a->split_k > 0 ? a->split_k : 1;
    if (split > nk) split = nk;
    if (a->epilogue != VP_EPI_SPLITK_F32) split = 1;
    const int ksplit_tiles = (nk + split - 1) / split;
    dim3 grid(ntiles, split);
    // tail handling (see gemm_kernel): only when a workspace is supplied and it pays
    Tail tl{ntiles, 1, nullptr};
    const int slots = 2 * num_cus();
    const int ntail = ntiles % slots;
    // measured on ViTPose-B shapes (12288 rows): the slab round trip + fix-up launch costs ~10-20 us,
    // which only pays for long reductions with the (slower) transposed-read Y operand, or on request
    const bool worth = a->tail_force || (YK && nk >= 32);
    if (worth && a->epilogue != VP_EPI_SPLITK_F32 && a->tail_ws && ntiles > slots && ntail > 0 &&
        ntail <= (slots * 3) / 4 && nk >= 4) {
        int ts = slots / ntail;
        if (ts > nk / 2) ts = nk / 2;  // at least two K steps per slice
        if (ts > 8) ts = 8;
        const size_t need = (size_t)ntail * ts * 16 * NTHREADS * sizeof(f32x4);
        if (ts >= 2 && need <= a->tail_ws_bytes) {
            tl = Tail{ntiles - ntail, ts, a->tail_ws};
            grid = dim3(tl.full + ntail * ts, 1);
        }
    }
    if (a->tile_mode == 4) {
        if (XK || split > 1) return VP_ERR_UNSUPPORTED;   // forward / dgrad products only
        const int tm4 = (a->M + W_BM - 1) / W_BM, tn4 = (a->N + W_BN - 1) / W_BN, nt4 = tm4 * tn4;
        if constexpr (!XK) {
#define VP_GEMM4_CASE(EPI)                                                                              \
    case EPI:                                                                                           \
        hipLaunchKernelGGL((gemm_w2_kernel<YK, EPI>), dim3(nt4), dim3(NTHREADS), 0, st, X, Y, a->M, \
                           a->N, a->K, tn4, nt4, e);                                                    \
        break;
            switch (a->epilogue) {
                VP_GEMM4_CASE(VP_EPI_BF16)
                VP_GEMM4_CASE(VP_EPI_GELU)
                VP_GEMM4_CASE(VP_EPI_RESID_F32)
                VP_GEMM4_CASE(VP_EPI_DGELU)
                VP_GEMM4_CASE(VP_EPI_F32)
                VP_GEMM4_CASE(VP_EPI_SCALE_RELU)
                VP_GEMM4_CASE(VP_EPI_DRELU)
                default:
                    return VP_ERR_ARG;
            }
#undef VP_GEMM4_CASE
        }
        VP_LAUNCH_RET();
    }
    if (a->gather16_C > 0) {
        // Y = reduction-major 16-tap gather of a->y (the deconv weight gradient): 256 x 256 tiles, split-K slabs
        if constexpr (XK && YK) {
            if (a->epilogue != VP_EPI_SPLITK_F32) return VP_ERR_UNSUPPORTED;
            const int tm2 = (a->M + T2 - 1) / T2, tn2 = (a->N + T2 - 1) / T2, nt2 = tm2 * tn2;
            Gather g{a->gather16_H, a->gather16_W, a->gather16_C, (const bf16*)a->gather16_zero, 1};
            hipLaunchKernelGGL((gemm256_kernel<true, true, VP_EPI_SPLITK_F32, true>), dim3(nt2 * split), dim3(T2_THREADS),
                               0, st, X, Y, a->M, a->N, a->K, tn2, nt2, ksplit_tiles, e, nullptr, 0, g);
            VP_LAUNCH_RET();
        } else {
            return VP_ERR_UNSUPPORTED;
        }
    }
    if (a->route_expert) {
        // per-image expert routing: 192-row tiles (one image each), forward / data-gradient products
        if (XK || split > 1 || a->M % T3 != 0 || a->route_split % 48 != 0 || !a->route_delta) return VP_ERR_UNSUPPORTED;
        const int tn3 = (a->N + T3 - 1) / T3, nt3 = (a->M / T3) * tn3;
        const Route rt{a->route_expert, a->route_delta, a->route_bias_delta, a->route_split};
        if constexpr (!XK) {
#define VP_GEMMR_CASE(EPI)                                                                                       \
    case EPI:                                                                                                    \
        hipLaunchKernelGGL((gemm192_kernel<false, YK, EPI, true>), dim3(nt3), dim3(T2_THREADS), 0, st, X, Y,    \
                           a->M, a->N, a->K, tn3, nt3, ksplit_tiles, e, nullptr, 0, rt);                         \
        break;
            switch (a->epilogue) {
                VP_GEMMR_CASE(VP_EPI_RESID_F32)
                VP_GEMMR_CASE(VP_EPI_F32)
                VP_GEMMR_CASE(VP_EPI_DGELU)
                VP_GEMMR_CASE(VP_EPI_BF16)
                default:
                    return VP_ERR_UNSUPPORTED;
            }
#undef VP_GEMMR_CASE
        }
        VP_LAUNCH_RET();
    }
    if (a->tile_mode == 5) {
        // persistent 192-tile kernel (epilogue of tile i under the K loop of tile i + 1): forward / data-gradient
        // products with a bf16 output, more than one round of tiles, at least 9 K tiles
        const int tm5 = (a->M + T3 - 1) / T3, tn5 = (a->N + T3 - 1) / T3, nt5 = tm5 * tn5;
        if (XK || split > 1 || nk < 10 || a->out_row_div > 0 || a->colscale || a->M % T3 != 0 || a->N < 4)
            return VP_ERR_UNSUPPORTED;
        const int grid5 = nt5 < num_cus() ? nt5 : num_cus();
        if constexpr (!XK) {
#define VP_GEMM5_CASE(EPI)                                                                                    \
    case EPI:                                                                                                 \
        hipLaunchKernelGGL((gemm192p_kernel<false, YK, EPI>), dim3(grid5), dim3(T2_THREADS), 0, st, X, Y, a->M, \
                           a->N, a->K, tn5, nt5, e);                                                          \
        break;
            switch (a->epilogue) {
                VP_GEMM5_CASE(VP_EPI_BF16)
                VP_GEMM5_CASE(VP_EPI_GELU)
                VP_GEMM5_CASE(VP_EPI_DGELU)
                default:
                    return VP_ERR_UNSUPPORTED;
            }
#undef VP_GEMM5_CASE
        }
        VP_LAUNCH_RET();
    }
    if (a->tile_mode == 6) {
        // 192 x 128 tiles of the 192-tile kernel (one workgroup per CU): forward / data-gradient products whose
        // 192-wide column tiles would leave CUs idle
        if (XK || split > 1) return VP_ERR_UNSUPPORTED;
        const int tm6 = (a->M + T3 - 1) / T3, tn6 = (a->N + 127) / 128, nt6 = tm6 * tn6;
        if constexpr (!XK) {
#define VP_GEMM6_CASE(EPI)                                                                                        \
    case EPI:                                                                                                     \
        hipLaunchKernelGGL((gemm192_kernel<false, YK, EPI, false, false, false, 2>), dim3(nt6), dim3(T2_THREADS), 0, \
                           st, X, Y, a->M, a->N, a->K, tn6, nt6, ksplit_tiles, e, nullptr, 0);                    \
        break;
            switch (a->epilogue) {
                VP_GEMM6_CASE(VP_EPI_BF16)
                VP_GEMM6_CASE(VP_EPI_GELU)
                VP_GEMM6_CASE(VP_EPI_RESID_F32)
                VP_GEMM6_CASE(VP_EPI_DGELU)
                VP_GEMM6_CASE(VP_EPI_F32)
                VP_GEMM6_CASE(VP_EPI_SCALE_RELU)
                VP_GEMM6_CASE(VP_EPI_DRELU)
                default:
                    return VP_ERR_ARG;
            }
#undef VP_GEMM6_CASE
        }
        VP_LAUNCH_RET();
    }
    if (a->tile_mode == 7) {
        // 192 x 192 tiles with the K tile split between the wave groups: long reductions, no K slices
        if (split > 1 || nk < 2) return VP_ERR_UNSUPPORTED;
        const int tm7 = (a->M + T3 - 1) / T3, tn7 = (a->N + T3 - 1) / T3, nt7 = tm7 * tn7;
#define VP_GEMM7_CASE(EPI)                                                                                   \
    case EPI:                                                                                                \
        hipLaunchKernelGGL((gemm192k_kernel<XK, YK, EPI>), dim3(nt7), dim3(T2_THREADS), 0, st, X, Y, a->M,  \
                           a->N, a->K, tn7, nt7, e, nullptr, 0);                                             \
        break;
        switch (a->epilogue) {
            VP_GEMM7_CASE(VP_EPI_BF16)
            VP_GEMM7_CASE(VP_EPI_RESID_F32)
            VP_GEMM7_CASE(VP_EPI_F32)
            default:
                return VP_ERR_UNSUPPORTED;
        }
#undef VP_GEMM7_CASE
        VP_LAUNCH_RET();
    }
    if (a->tile_mode == 3) {
        const int tm3 = (a->M + T3 - 1) / T3, tn3 = (a->N + T3 - 1) / T3, nt3 = tm3 * tn3;
        dim3 grid3(nt3 * split);
#define VP_GEMM3_CASE(EPI)                                                                             \
    case EPI:                                                                                          \
        hipLaunchKernelGGL((gemm192_kernel<XK, YK, EPI>), grid3, dim3(T2_THREADS), 0, st, X, Y, a->M, \
                           a->N, a->K, tn3, nt3, ksplit_tiles, e, nullptr, 0);                                     \
        break;
        switch (a->epilogue) {
            VP_GEMM3_CASE(VP_EPI_BF16)
            VP_GEMM3_CASE(VP_EPI_GELU)
            VP_GEMM3_CASE(VP_EPI_RESID_F32)
            VP_GEMM3_CASE(VP_EPI_DGELU)
            VP_GEMM3_CASE(VP_EPI_SPLITK_F32)
            VP_GEMM3_CASE(VP_EPI_F32)
            VP_GEMM3_CASE(VP_EPI_SCALE_RELU)
            VP_GEMM3_CASE(VP_EPI_DRELU)
            default:
                return VP_ERR_ARG;
        }
#undef VP_GEMM3_CASE
        VP_LAUNCH_RET();
    }
    // 256 x 256 ping-pong kernel: large problems whose tile count fills the chip
    const int tm2 = (a->M + T2 - 1) / T2, tn2 = (a->N + T2 - 1) / T2;
    bool big = a->tile_mode == 2;
    if (a->tile_mode == 0) big = use_big_tiles(a->M, a->N, a->K, tm2 * tn2 * split, num_cus());
    if (big) {
        const int nt2 = tm2 * tn2;
        dim3 grid2(nt2 * split);
#define VP_GEMM2_CASE(EPI)                                                                             \
    case EPI:                                                                                          \
        hipLaunchKernelGGL((gemm256_kernel<XK, YK, EPI>), grid2, dim3(T2_THREADS), 0, st, X, Y, a->M, \
                           a->N, a->K, tn2, nt2, ksplit_tiles, e, nullptr, 0);                                     \
        break;
        switch (a->epilogue) {
            VP_GEMM2_CASE(VP_EPI_BF16)
            VP_GEMM2_CASE(VP_EPI_GELU)
            VP_GEMM2_CASE(VP_EPI_RESID_F32)
            VP_GEMM2_CASE(VP_EPI_DGELU)
            VP_GEMM2_CASE(VP_EPI_SPLITK_F32)
            VP_GEMM2_CASE(VP_EPI_F32)
            VP_GEMM2_CASE(VP_EPI_SCALE_RELU)
            VP_GEMM2_CASE(VP_EPI_DRELU)
            default:
                return VP_ERR_ARG;
        }
#undef VP_GEMM2_CASE
        VP_LAUNCH_RET();
    }
#define VP_GEMM_CASE(EPI)                                                                         \
    case EPI:                                                                                     \
        hipLaunchKernelGGL((gemm_kernel<XK, YK, EPI>), grid, dim3(NTHREADS), 0, st, X, Y, a->M,   \
                           a->N, a->K, tiles_n, ntiles, ksplit_tiles, e, tl);                     \
        if (tl.split > 1)                                                                         \
            hipLaunchKernelGGL((gemm_tail_fixup<EPI>), dim3(ntiles - tl.full), dim3(NTHREADS), 0, \
                               st, a->M, a->N, tiles_n, ntiles, e, tl);                           \
        break;
    switch (a->epilogue) {
        VP_GEMM_CASE(VP_EPI_BF16)
        VP_GEMM_CASE(VP_EPI_GELU)
        VP_GEMM_CASE(VP_EPI_RESID_F32)
        VP_GEMM_CASE(VP_EPI_DGELU)
        VP_GEMM_CASE(VP_EPI_SPLITK_F32)
        VP_GEMM_CASE(VP_EPI_F32)
        VP_GEMM_CASE(VP_EPI_SCALE_RELU)
        VP_GEMM_CASE(VP_EPI_DRELU)
        default:
            return VP_ERR_ARG;
    }
#undef VP_GEMM_CASE
    VP_LAUNCH_RET();
}

// hi / lo pair operands (vp_gemm_args.split3): k-contiguous X and Y only, epilogues F32 / RESID_F32 / PAIR; the
// kernels see a reduction extent of 2 K pair elements.
int launch_split3(const vp_gemm_args* a, hipStream_t st) {
    const int K2 = 2 * a->K;
    Operand X{(const bf16*)a->x, a->ldx, a->M}, Y{(const bf16*)a->y, a->ldy, a->N};
    Epi e{a->epilogue, a->out, a->ldo, a->out2, a->bias, a->resid, a->rowscale, a->rows_per_scale,
          a->aux, a->ldaux, a->pos, a->pos_rows, a->colscale,
          a->out_row_div, a->out_row_hi, a->out_row_lo, a->out_row_off, a->split_stride, a->pair_act};
#define VP_S3_SWITCH(LAUNCH)                                  \
    switch (a->epilogue) {                                    \
        case VP_EPI_F32: LAUNCH(VP_EPI_F32); break;           \
        case VP_EPI_RESID_F32: LAUNCH(VP_EPI_RESID_F32); break; \
        case VP_EPI_PAIR: LAUNCH(VP_EPI_PAIR); break;         \
        default: return VP_ERR_UNSUPPORTED;                   \
    }
    if (a->route_expert) {
        if (a->M % T3 != 0 || a->route_split % 48 != 0 || !a->route_delta) return VP_ERR_UNSUPPORTED;
        if (a->epilogue != VP_EPI_RESID_F32) return VP_ERR_UNSUPPORTED;
        const int tn3 = (a->N + T3 - 1) / T3, nt3 = (a->M / T3) * tn3;
        const Route rt{a->route_expert, a->route_delta, a->route_bias_delta, a->route_split};
        hipLaunchKernelGGL((gemm192_kernel<false, false, VP_EPI_RESID_F32, true, false, true>), dim3(nt3), dim3(T2_THREADS),
                           0, st, X, Y, a->M, a->N, K2, tn3, nt3, K2 / BK, e, nullptr, 0, rt);
        VP_LAUNCH_RET();
    }
    int mode = a->tile_mode;
    if (mode == 0) {
        const int tm2 = (a->M + T2 - 1) / T2, tn2 = (a->N + T2 - 1) / T2;
        mode = use_big_tiles(a->M, a->N, K2, tm2 * tn2, num_cus()) ? 2 : 1;
        if (mode == 1 && a->M % T3 == 0 && a->N % T3 == 0 && a->M >= 2 * T3) mode = 3;
    }
    if (mode == 4) {
        const int tm = (a->M + W_BM - 1) / W_BM, tn = (a->N + W_BN - 1) / W_BN, nt = tm * tn;
#define VP_S3_W2(EPI)                                                                                          \
    hipLaunchKernelGGL((gemm_w2_kernel<false, EPI, false, true>), dim3(nt), dim3(NTHREADS), 0, st, X, Y, a->M, \
                       a->N, K2, tn, nt, e)
        VP_S3_SWITCH(VP_S3_W2)
#undef VP_S3_W2
    } else if (mode == 3) {
        const int tm = (a->M + T3 - 1) / T3, tn = (a->N + T3 - 1) / T3, nt = tm * tn;
#define VP_S3_192(EPI)                                                                                              \
    hipLaunchKernelGGL((gemm192_kernel<false, false, EPI, false, false, true>), dim3(nt), dim3(T2_THREADS), 0, st, \
                       X, Y, a->M, a->N, K2, tn, nt, K2 / BK, e, nullptr, 0)
        VP_S3_SWITCH(VP_S3_192)
#undef VP_S3_192
    } else if (mode == 2) {
        const int tm = (a->M + T2 - 1) / T2, tn = (a->N + T2 - 1) / T2, nt = tm * tn;
#define VP_S3_256(EPI)                                                                                          \
    hipLaunchKernelGGL((gemm256_kernel<false, false, EPI, false, true>), dim3(nt), dim3(T2_THREADS), 0, st, X, \
                       Y, a->M, a->N, K2, tn, nt, K2 / BK, e, nullptr, 0)
        VP_S3_SWITCH(VP_S3_256)
#undef VP_S3_256
    } else {
        const int tm = (a->M + BM - 1) / BM, tn = (a->N + BN - 1) / BN, nt = tm * tn;
        const Tail tl{nt, 1, nullptr};
#define VP_S3_128(EPI)                                                                                         \
    hipLaunchKernelGGL((gemm_kernel<false, false, EPI, true>), dim3(nt, 1), dim3(NTHREADS), 0, st, X, Y, a->M, \
                       a->N, K2, tn, nt, K2 / BK, e, tl)
        VP_S3_SWITCH(VP_S3_128)
#undef VP_S3_128
    }
#undef VP_S3_SWITCH
    VP_LAUNCH_RET();
}

}  // namespace

extern "C" int vp_gemm_ln_slots(int N, int tile_mode) { return ln_slots_of(N, tile_mode); }

extern "C" int vp_gemm_bf16(const vp_gemm_args* a, hipStream_t stream) {
    VP_CHECK_ARG(a && a->x && a->y && a->out);
    VP_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0);
    if (a->split3) {
        // hi / lo pair operands: 32 values per 64-element K step; rows of 2 K pair elements
        VP_CHECK_ARG(a->K % 32 == 0 && a->N % 4 == 0 && !a->x_kmajor && !a->y_kmajor && a->gather16_C == 0);
        VP_CHECK_ARG(a->ldx % 8 == 0 && a->ldy % 8 == 0 && a->ldo % 4 == 0 && a->ldx >= 2 * a->K && a->ldy >= 2 * a->K);
        VP_CHECK_ARG(((uintptr_t)a->x & 15) == 0 && ((uintptr_t)a->y & 15) == 0 && ((uintptr_t)a->out & 15) == 0);
        if (a->epilogue == VP_EPI_PAIR)
            VP_CHECK_ARG(a->N % 32 == 0 && a->ldo >= 2 * a->N && a->pair_act >= 0 && a->pair_act <= 2);
        if (a->rowscale) VP_CHECK_ARG(a->rows_per_scale > 0);
        if (a->pos) VP_CHECK_ARG(a->pos_rows > 0);
        return launch_split3(a, stream);
    }
    if (a->epilogue == VP_EPI_PAIR) return VP_ERR_UNSUPPORTED;   // pair outputs come from pair operands
    VP_CHECK_ARG(a->K % BK == 0);          // reduction extent: whole 64-wide steps
    VP_CHECK_ARG(a->N % 4 == 0);           // epilogue handles 4 columns per lane
    VP_CHECK_ARG(a->ldx % 8 == 0 && a->ldy % 8 == 0 && a->ldo % 4 == 0);
    VP_CHECK_ARG(((uintptr_t)a->x & 15) == 0 && ((uintptr_t)a->y & 15) == 0 && ((uintptr_t)a->out & 15) == 0);
    if (a->x_kmajor) VP_CHECK_ARG(a->M % 8 == 0 && a->M >= 8);
    if (a->y_kmajor) VP_CHECK_ARG(a->N % 8 == 0 && a->N >= 8);
    if (a->epilogue == VP_EPI_DGELU || a->epilogue == VP_EPI_DRELU) VP_CHECK_ARG(a->aux && a->ldaux % 4 == 0);
    if (a->out2) VP_CHECK_ARG(((uintptr_t)a->out2 & 15) == 0);
    if (a->rowscale) VP_CHECK_ARG(a->rows_per_scale > 0);
    if (a->pos) VP_CHECK_ARG(a->pos_rows > 0);
    if (a->route_expert) VP_CHECK_ARG(a->route_delta && a->route_split >= 0 && !a->x_kmajor);
    if (a->gather16_C > 0) {
        const int H = a->gather16_H, W = a->gather16_W, C = a->gather16_C;
        VP_CHECK_ARG(a->gather16_zero && H > 0 && W > 0 && W <= BK && C % 128 == 0 && a->N == 16 * C);
        VP_CHECK_ARG(a->K % (H * W) == 0 && BK / W + 1 <= H);             // whole images; one row wrap per K tile
        VP_CHECK_ARG((size_t)a->K * 4 < (size_t)0x7fffffff);               // pixel indices are 32-bit
    }
    if (a->x_kmajor) return a->y_kmajor ? launch_layout<true, true>(a, stream) : launch_layout<true, false>(a, stream);
    return a->y_kmajor ? launch_layout<false, true>(a, stream) : launch_layout<false, false>(a, stream);
}

static int group_tile(int tile_mode) { return tile_mode == 2 ? T2 : T3; }

#if defined(VP_CLOCK_PROBE) && VP_CLOCK_PROBE == 2
// probe build only: the per-workgroup records of the last launch of an instrumented kernel
extern "C" int vp_probe_dump(long long* host, int words) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vp_tile_log), (size_t)words * 8);
}
#endif

extern "C" int vp_wgrad_group_table(vp_group_entry* problems, int count, int tile_mode, int* total_tiles) {
    VP_CHECK_ARG(problems && total_tiles && count > 0 && count <= 64 && (tile_mode == 2 || tile_mode == 3 || tile_mode == 7));
    const int T = group_tile(tile_mode);
    int t0 = 0;
    for (int i = 0; i < count; ++i) {
        vp_group_entry& g = problems[i];
        VP_CHECK_ARG(g.x && g.y && g.out && g.M >= 8 && g.N >= 8 && g.M % 8 == 0 && g.N % 8 == 0);
        VP_CHECK_ARG(g.ldx % 8 == 0 && g.ldy % 8 == 0 && g.ldo % 4 == 0 && g.ldx >= g.M && g.ldy >= g.N && g.ldo >= g.N);
        VP_CHECK_ARG(((uintptr_t)g.x & 15) == 0 && ((uintptr_t)g.y & 15) == 0 && ((uintptr_t)g.out & 15) == 0);
        if (g.image_mask) VP_CHECK_ARG(tile_mode == 3 && g.tiles_per_image >= 1);   // masked problems: 192-tile kernel
        g.tiles_n = (g.N + T - 1) / T;
        g.ntiles = ((g.M + T - 1) / T) * g.tiles_n;
        g.tile0 = t0;
        t0 += g.ntiles;
    }
    *total_tiles = t0;
    return VP_OK;
}

extern "C" int vp_wgrad_grouped_bf16(const vp_group_entry* table_dev, int count, int total_tiles, int K,
                                     int tile_mode, hipStream_t stream) {
    VP_CHECK_ARG(table_dev && count > 0 && count <= 64 && total_tiles > 0 && K > 0 && K % BK == 0);
    VP_CHECK_ARG(tile_mode == 2 || tile_mode == 3 || tile_mode == 7);
    Operand none{nullptr, 0, 0};
    Epi e{};
    e.kind = VP_EPI_F32;
    if (tile_mode == 7)
        hipLaunchKernelGGL((gemm192k_kernel<true, true, VP_EPI_F32>), dim3(total_tiles), dim3(T2_THREADS), 0, stream,
                           none, none, 0, 0, K, 1, 1, e, table_dev, count);
    else if (tile_mode == 2)
        hipLaunchKernelGGL((gemm256_kernel<true, true, VP_EPI_F32>), dim3(total_tiles), dim3(T2_THREADS), 0, stream,
                           none, none, 0, 0, K, 1, 1, K / BK, e, table_dev, count);
    else
        hipLaunchKernelGGL((gemm192_kernel<true, true, VP_EPI_F32>), dim3(total_tiles), dim3(T2_THREADS), 0, stream,
                           none, none, 0, 0, K, 1, 1, K / BK, e, table_dev, count);
    VP_LAUNCH_RET();
}

extern "C" int vp_deconv_dgrad_bf16(const void* dout, const void* wd, void* din, int B, int H, int W, int C, int Cin,
                                    const void* relu_aux, const void* zero_page, hipStream_t stream) {
    VP_CHECK_ARG(dout && wd && din && zero_page && B > 0 && H > 0 && W > 0);
    VP_CHECK_ARG(C % 64 == 0 && Cin % 8 == 0);
    VP_CHECK_ARG(((uintptr_t)dout & 15) == 0 && ((uintptr_t)wd & 15) == 0 && ((uintptr_t)din & 15) == 0);
    VP_CHECK_ARG((size_t)B * 4 * H * W < (size_t)0x7fffffff / 2);      // pixel indices are 32-bit
    const int M = B * H * W, K = 16 * C;
    Operand X{(const bf16*)dout, C, M}, Y{(const bf16*)wd, K, Cin};
    Epi e{};
    e.kind = relu_aux ? VP_EPI_DRELU : VP_EPI_BF16;
    e.out = din;
    e.ldo = Cin;
    e.aux = relu_aux;
    e.ldaux = Cin;
    Gather g{H, W, C, (const bf16*)zero_page, 1};
    const int tm = (M + W_BM - 1) / W_BM, tn = (Cin + W_BN - 1) / W_BN;
    if (relu_aux)
        hipLaunchKernelGGL((gemm_w2_kernel<false, VP_EPI_DRELU, true>), dim3(tm * tn), dim3(NTHREADS), 0, stream, X, Y, M,
                           Cin, K, tn, tm * tn, e, g);
    else
        hipLaunchKernelGGL((gemm_w2_kernel<false, VP_EPI_BF16, true>), dim3(tm * tn), dim3(NTHREADS), 0, stream, X, Y, M,
                           Cin, K, tn, tm * tn, e, g);
    VP_LAUNCH_RET();
}

extern "C" int vp_wgrad_grouped_kmap_bf16(const vp_group_entry* table_dev, int count, int total_tiles, int K,
                                          hipStream_t stream) {
    VP_CHECK_ARG(table_dev && count > 0 && count <= 64 && total_tiles > 0 && K > 0 && K % BK == 0);
    Operand none{nullptr, 0, 0};
    Epi e{};
    e.kind = VP_EPI_F32;
    hipLaunchKernelGGL((gemm192_kernel<true, true, VP_EPI_F32, false, true>), dim3(total_tiles), dim3(T2_THREADS), 0,
                       stream, none, none, 0, 0, K, 1, 1, K / BK, e, table_dev, count,
                       Route{nullptr, nullptr, nullptr, 0});
    VP_LAUNCH_RET();
}

extern "C" int vp_deconv_fwd_bf16(const void* in, const void* wph, void* out, int B, int H, int W, int C, int Cout,
                                  int epilogue, const float* colscale, const float* shift, const void* zero_page,
                                  hipStream_t stream) {
    VP_CHECK_ARG(in && wph && out && zero_page && B > 0 && H > 0 && W > 0);
    VP_CHECK_ARG(C % 64 == 0 && Cout % 8 == 0 && (epilogue == VP_EPI_BF16 || epilogue == VP_EPI_SCALE_RELU));
    VP_CHECK_ARG(((uintptr_t)in & 15) == 0 && ((uintptr_t)wph & 15) == 0 && ((uintptr_t)out & 15) == 0);
    const int M = B * H * W, K = 4 * C;
    Operand X{(const bf16*)in, C, M}, Y{(const bf16*)wph, K, Cout};
    Epi e{};
    e.kind = epilogue;
    e.out = out;
    e.ldo = Cout;
    e.bias = shift;
    e.colscale = colscale;
    Gather g{H, W, C, (const bf16*)zero_page, 0};
    const int tm = (M + W_BM - 1) / W_BM, tn = (Cout + W_BN - 1) / W_BN;
    dim3 grid(tm * tn, 4);
    if (epilogue == VP_EPI_BF16)
        hipLaunchKernelGGL((gemm_w2_kernel<false, VP_EPI_BF16, true>), grid, dim3(NTHREADS), 0, stream, X, Y, M, Cout, K,
                           tn, tm * tn, e, g);
    else
        hipLaunchKernelGGL((gemm_w2_kernel<false, VP_EPI_SCALE_RELU, true>), grid, dim3(NTHREADS), 0, stream, X, Y, M,
                           Cout, K, tn, tm * tn, e, g);
    VP_LAUNCH_RET();
}

extern "C" int vp_deconv_fwd_split(const void* in_pair, const void* wph_pair, void* out_pair, int B, int H, int W, int C,
                                   int Cout, const float* colscale, const float* shift, const void* zero_page,
                                   hipStream_t stream) {
    VP_CHECK_ARG(in_pair && wph_pair && out_pair && zero_page && B > 0 && H > 0 && W > 0);
    VP_CHECK_ARG(C % 32 == 0 && Cout % 32 == 0);
    VP_CHECK_ARG(((uintptr_t)in_pair & 15) == 0 && ((uintptr_t)wph_pair & 15) == 0 && ((uintptr_t)out_pair & 15) == 0);
    // the plain implicit-GEMM loader on pair rows: a pixel has 2 C pair elements, a tap 2 C reduction steps
    const int M = B * H * W, C2 = 2 * C, K2 = 4 * C2;
    Operand X{(const bf16*)in_pair, C2, M}, Y{(const bf16*)wph_pair, K2, Cout};
    Epi e{};
    e.kind = VP_EPI_PAIR;
    e.out = out_pair;
    e.ldo = 2 * Cout;
    e.bias = shift;
    e.colscale = colscale;
    e.pair_act = 2;
    Gather g{H, W, C2, (const bf16*)zero_page, 0};
    const int tm = (M + W_BM - 1) / W_BM, tn = (Cout + W_BN - 1) / W_BN;
    hipLaunchKernelGGL((gemm_w2_kernel<false, VP_EPI_PAIR, true, true>), dim3(tm * tn, 4), dim3(NTHREADS), 0, stream, X, Y,
                       M, Cout, K2, tn, tm * tn, e, g);
    VP_LAUNCH_RET();
}
