// LayerNorm forward / backward over the channel dimension of token-major activations.
// One wavefront per row: the row lives in registers (float4 chunks strided by 64 lanes),
// statistics are fp32 wave reductions, every global access is 8 or 16 bytes per lane.
// HBM-bound: forward moves 4E (x) + 2E (y) bytes per row; backward 4E+2E in, 4E+2E out.
#include <mutex>

#include "common.h"
#include "vitpose_hip.h"

// Partial column sums of vp_layernorm_bwd_static: [512 workgroups][3][E <= 1280] floats (7.5 MiB), part of the
// code object, so every GPU the library runs on holds one copy until the process ends.  Host code never uses
// this name as an address: static_scratch() asks the runtime for the device address once per device.
static __device__ __align__(256) float scratch[VP_LN_STATIC_MAX_BLOCKS * 3 * VP_LN_STATIC_MAX_E];

namespace {

// CH = float4 chunks per lane (row length E <= 256 * CH); the row loops are fully unrolled so the
// register footprint follows the actual row length (E = 768 -> 3 chunks) and occupancy stays high.
// PAIR: y is a hi / lo pair array (rows of 2 E bf16: groups of 32 hi values followed by their 32 lo values,
// hi = bf16(v), lo = bf16(v - hi)) of max(v, 0) when `relu` is set, else of v: the operand of an fp32-accurate GEMM.
template <int CH, bool PAIR, int R>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16* __restrict__ y,
                                                     float* __restrict__ y32, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int M, int E, float eps,
                                                     int relu = 0) {
    const int lane = threadIdx.x & 63;
    const int nch = E >> 2;
    const int nwaves = gridDim.x * 4;
    const float invE = 1.f / (float)E;
    // affine coefficients once per wave (the wave walks several rows)
    f32x4 g[CH], b[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < nch;
        g[i] = ok ? *reinterpret_cast<const f32x4*>(gamma + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
        b[i] = ok ? *reinterpret_cast<const f32x4*>(beta + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // A wave walks groups of R rows (w + r W, r < R; then w + (R + r) W, ...; W = waves of the grid) through TWO
    // register buffers: while group p is reduced, normalised and stored, the loads of group p + 1 are in flight, and
    // group p + 2 is requested as soon as p's registers are free -- the reads and writes of a wave overlap, and so do
    // those of the waves of a CU, instead of every wave of the chip loading and then every wave storing.  R = 2 rows
    // per group where the registers allow it (rows of <= 1280 channels), else 1.
    auto load_group = [&](int row0, f32x4 (&v)[R][CH]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r * nwaves;
            const float* xr = x + (size_t)(row < M ? row : row0) * E;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int c = lane + 64 * i;
                v[r][i] = c < nch ? *reinterpret_cast<const f32x4*>(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto process_group = [&](int row0, f32x4 (&v)[R][CH]) {
        // Every row must get the same arithmetic whichever slot r of a group it lands in (an image's result may not
        // depend on the size of the batch around it): no contraction left to the compiler's choice -- it had fused
        // the squares of one row of a pair and not of the other -- the fused operations are written out.
#pragma clang fp contract(off)
        float s[R], q[R], mu[R], rs[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s[r] = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) s[r] += v[r][i][0] + v[r][i][1] + v[r][i][2] + v[r][i][3];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) mu[r] = wave_sum(s[r]) * invE;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            q[r] = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (lane + 64 * i < nch) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = v[r][i][e] - mu[r];
                        q[r] = __builtin_fmaf(d, d, q[r]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) rs[r] = rsqrtf(wave_sum(q[r]) * invE + eps);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r * nwaves;
            if (row >= M) break;
            if (lane == 0) {
                if (mean_out) mean_out[row] = mu[r];
                if (rstd_out) rstd_out[row] = rs[r];
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf((v[r][i][e] - mu[r]) * rs[r], g[i][e], b[i][e]);
                    if constexpr (PAIR) {
                        if (relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                        }
                        const int pc = ((c >> 3) << 6) | ((c & 7) << 2);     // column 4c of the row in the pair layout
                        const bf16x4 h = pack4(o[0], o[1], o[2], o[3]);
                        bf16* p = y + (size_t)row * 2 * E + pc;
                        *reinterpret_cast<bf16x4*>(p) = h;
                        *reinterpret_cast<bf16x4*>(p + 32) = pack4(o[0] - (float)h[0], o[1] - (float)h[1],
                                                                  o[2] - (float)h[2], o[3] - (float)h[3]);
                    } else if (y) {
                        *reinterpret_cast<bf16x4*>(y + (size_t)row * E + 4 * c) = pack4(o[0], o[1], o[2], o[3]);
                    }
                    if (y32) *reinterpret_cast<f32x4*>(y32 + (size_t)row * E + 4 * c) = o;
                }
            }
        }
    };
    f32x4 bufA[R][CH], bufB[R][CH];
    int rowA = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rowA >= M) return;
    int rowB = rowA + R * nwaves;
    load_group(rowA, bufA);
    if (rowB < M) load_group(rowB, bufB);
    while (true) {
        process_group(rowA, bufA);
        rowA += 2 * R * nwaves;
        if (rowA < M) load_group(rowA, bufA);
        if (rowB >= M) break;
        process_group(rowB, bufB);
        rowB += 2 * R * nwaves;
        if (rowB < M) load_group(rowB, bufB);
        if (rowA >= M) break;
    }
}

// Backward.  512-thread workgroups; each wave walks rows w, w + W, ... and keeps per-column partial
// sums of dgamma / dbeta / (scaled dx column sum) in registers; one LDS reduction over the block's
// 8 waves, then the block stores its partial sums for the finalize kernel.
template <int CH>
__global__ __launch_bounds__(512) void ln_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ dx, bf16* __restrict__ dxb,
                                                     const float* __restrict__ rowscale, int rows_per_scale,
                                                     float* __restrict__ partial, int M, int E) {
    __shared__ float red[3][8][64 * 4 + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = E >> 2;
    const int nwaves = gridDim.x * 8;
    const float invE = 1.f / (float)E;
    f32x4 ag[CH], ab[CH], ac[CH], g[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        ag[i] = ab[i] = ac[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = lane + 64 * i;
        g[i] = c < nch ? *reinterpret_cast<const f32x4*>(gamma + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row = blockIdx.x * 8 + wave; row < M; row += nwaves) {
        // every load of the row is issued before the first use
        f32x4 xv[CH], dr[CH];
        bf16x4 dv[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = lane + 64 * i;
            const bool ok = c < nch;
            const size_t o = (size_t)row * E + 4 * (ok ? c : 0);
            xv[i] = *reinterpret_cast<const f32x4*>(x + o);
            dv[i] = *reinterpret_cast<const bf16x4*>(dy + o);
            dr[i] = dres ? *reinterpret_cast<const f32x4*>(dres + o) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float mu = mean[row], rs = rstd[row];
        const float sc = rowscale ? rowscale[row / rows_per_scale] : 1.f;
        f32x4 xh[CH], gy[CH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (lane + 64 * i < nch) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = (float)dv[i][r];
                    xh[i][r] = (xv[i][r] - mu) * rs;
                    gy[i][r] = d * g[i][r];
                    s1 += gy[i][r];
                    s2 += gy[i][r] * xh[i][r];
                    ag[i][r] += d * xh[i][r];
                    ab[i][r] += d;
                }
            }
        }
        s1 = wave_sum(s1) * invE;
        s2 = wave_sum(s2) * invE;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = rs * (gy[i][r] - s1 - xh[i][r] * s2) + dr[i][r];
                if (dx) *reinterpret_cast<f32x4*>(dx + (size_t)row * E + 4 * c) = o;
                o *= sc;
                ac[i] += o;
                if (dxb) *reinterpret_cast<bf16x4*>(dxb + (size_t)row * E + 4 * c) = pack4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    // cross-wave reduction, one chunk index at a time
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[0][wave][lane * 4 + r] = ag[i][r];
            red[1][wave][lane * 4 + r] = ab[i][r];
            red[2][wave][lane * 4 + r] = ac[i][r];
        }
        __syncthreads();
        // threads 0-255 cover the 256 columns of this chunk group; 256-511 idle here
        const int col = threadIdx.x;            // column within the group: lane' * 4 + r
        const int c = (col >> 2) + 64 * i;      // chunk index
        if (col < 256 && c < nch) {
            const int e = 4 * c + (col & 3);
            float g_ = 0.f, b_ = 0.f, c_ = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                g_ += red[0][w][col];
                b_ += red[1][w][col];
                c_ += red[2][w][col];
            }
            // per-block partial sums [block][3][E]; hundreds of blocks adding atomically into the
            // same E addresses would serialise at the memory side
            float* pp = partial + (size_t)blockIdx.x * 3 * E;
            pp[e] = g_;
            pp[E + e] = b_;
            pp[2 * E + e] = c_;
        }
    }
}

// sum the per-block partials over blocks and accumulate into dgamma / dbeta / dcolsum:
// 16 columns x 16 block lanes per workgroup, so a 3E-column table is covered by 3E/16 workgroups
__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int E,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ dcolsum) {
    __shared__ float red[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + cx;                      // index into the 3E partial columns
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < 3 * E) {
        int b = ry;
        for (; b + 48 < nblocks; b += 64) {
            s0 += partial[(size_t)b * 3 * E + j];
            s1 += partial[(size_t)(b + 16) * 3 * E + j];
            s2 += partial[(size_t)(b + 32) * 3 * E + j];
            s3 += partial[(size_t)(b + 48) * 3 * E + j];
        }
        for (; b < nblocks; b += 16) s0 += partial[(size_t)b * 3 * E + j];
    }
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && j < 3 * E) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][cx];
        float* dst = j < E ? dgamma : (j < 2 * E ? dbeta : dcolsum);
        if (dst) dst[j % E] += t;
    }
}

// the same reduction for several LayerNorm backward passes at once: blockIdx.y = entry of the table
__global__ __launch_bounds__(256) void ln_bwd_finalize_batched_kernel(const vp_ln_partial* __restrict__ table, int E) {
    __shared__ float red[16][17];
    const vp_ln_partial t = table[blockIdx.y];
    const float* partial = t.partial;
    const int nblocks = t.blocks;
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < 3 * E) {
        int b = ry;
        for (; b + 48 < nblocks; b += 64) {
            s0 += partial[(size_t)b * 3 * E + j];
            s1 += partial[(size_t)(b + 16) * 3 * E + j];
            s2 += partial[(size_t)(b + 32) * 3 * E + j];
            s3 += partial[(size_t)(b + 48) * 3 * E + j];
        }
        for (; b < nblocks; b += 16) s0 += partial[(size_t)b * 3 * E + j];
    }
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && j < 3 * E) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += red[r][cx];
        float* dst = j < E ? t.dgamma : (j < 2 * E ? t.dbeta : t.dcolsum);
        if (dst) dst[j % E] = t.overwrite ? acc : dst[j % E] + acc;
    }
}

// Grid of the backward kernel: one 512-thread workgroup per CU.  From 768 channels on a wave needs more than 128 VGPRs,
// so a CU holds one workgroup at a time and 512 workgroups ran as two rounds with twice the partial sums to reduce
// afterwards (measured inside the training step, 512 against 256 workgroups: ViTPose-B 10.46 -> 10.40 ms, ViTPose-L
// 17.64 -> 17.42 ms; ViTPose-S, whose waves fit twice, 5.545 either way).
inline int ln_bwd_blocks(int M, int E, size_t workspace_floats) {
    int blocks = (M + 7) / 8;
    if (blocks > 256) blocks = 256;
    if ((size_t)blocks * 3 * E > workspace_floats) blocks = (int)(workspace_floats / (3 * (size_t)E));
    return blocks;
}

// device address of `scratch` on the current device (looked up on the first call per device, then cached)
int static_scratch(float** out) {
    constexpr int MAX_DEVICES = 64;
    static float* ptr[MAX_DEVICES] = {};
    static std::mutex mu;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MAX_DEVICES) return VP_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!p) return VP_ERR_UNSUPPORTED;
        ptr[dev] = static_cast<float*>(p);
    }
    *out = ptr[dev];
    return VP_OK;
}

// Folded weights of a Linear behind a LayerNorm: one workgroup per output row n.
__global__ __launch_bounds__(256) void ln_fold_weights_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ bias,
                                                              bf16* __restrict__ Wg, float* __restrict__ s_out,
                                                              float* __restrict__ c_out, int K) {
    __shared__ float red[2][4];
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* w = W + (size_t)n * K;
    float s = 0.f, c = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float wk = w[k];
        const bf16 r = (bf16)(wk * gamma[k]);
        Wg[(size_t)n * K + k] = r;
        s += (float)r;
        c = __builtin_fmaf(wk, beta[k], c);
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if (lane == 0) {
        red[0][wave] = s;
        red[1][wave] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s_out[n] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        c_out[n] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]) + (bias ? bias[n] : 0.f);
    }
}

}  // namespace

extern "C" int vp_layernorm_bwd_blocks(int M, int E, size_t workspace_floats) {
    if (M <= 0 || E <= 0) return VP_ERR_ARG;
    return ln_bwd_blocks(M, E, workspace_floats);
}

// Grid of the forward kernel for groups of R rows: at most the cap, and such that every wave gets the same number
// of groups (6144 rows in single-row groups on 1024 workgroups would give half of the waves two rows, half one).
static int ln_fwd_blocks(int M, int R) {
    const int groups = (M + R - 1) / R;
    const int cap = 1024;       // 4 resident workgroups per CU (measured: 768 and 1024 level, 512 and 1536 slower)
    const int iters = (groups + 4 * cap - 1) / (4 * cap);
    const int blocks = (groups + 4 * iters - 1) / (4 * iters);
    return blocks < 1 ? 1 : blocks;
}

extern "C" int vp_layernorm_bwd_finalize(const vp_ln_partial* table_dev, int count, int E, hipStream_t stream) {
    VP_CHECK_ARG(table_dev && count > 0 && count <= 65535 && E > 0);
    hipLaunchKernelGGL(ln_bwd_finalize_batched_kernel, dim3((3 * E + 15) / 16, count), dim3(256), 0, stream, table_dev, E);
    VP_LAUNCH_RET();
}

extern "C" int vp_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                                float* mean, float* rstd, int M, int E, float eps, hipStream_t stream) {
    VP_CHECK_ARG(x && gamma && beta && (y_bf16 || y_f32));
    VP_CHECK_ARG(M > 0 && E > 0 && E % 4 == 0 && E <= 256 * 8);
    const int CHs = (E + 255) / 256;
    const bool two = CHs <= 5;             // two rows per register buffer up to 1280 channels (measured faster at every ViTPose width)
    const int blocks = ln_fwd_blocks(M, two ? 2 : 1);
#define VP_LNF(CH)                                                                                                  \
    if (two && CH <= 5)                                                                                             \
        hipLaunchKernelGGL((ln_fwd_kernel<CH, false, (CH <= 5 ? 2 : 1)>), dim3(blocks), dim3(256), 0, stream, x, gamma, beta, \
                           (bf16*)y_bf16, y_f32, mean, rstd, M, E, eps, 0);                                        \
    else                                                                                                            \
        hipLaunchKernelGGL((ln_fwd_kernel<CH, false, 1>), dim3(blocks), dim3(256), 0, stream, x, gamma, beta,      \
                           (bf16*)y_bf16, y_f32, mean, rstd, M, E, eps, 0)
    switch ((E + 255) / 256) {
        case 1: VP_LNF(1); break;
        case 2: VP_LNF(2); break;
        case 3: VP_LNF(3); break;
        case 4: VP_LNF(4); break;
        case 5: VP_LNF(5); break;
        case 6: VP_LNF(6); break;
        case 7: VP_LNF(7); break;
        default: VP_LNF(8); break;
    }
#undef VP_LNF
    VP_LAUNCH_RET();
}

extern "C" int vp_layernorm_fwd_pair(const float* x, const float* gamma, const float* beta, void* y_pair, float* y_f32,
                                     int relu, int M, int E, float eps, hipStream_t stream) {
    VP_CHECK_ARG(x && gamma && beta && y_pair);
    VP_CHECK_ARG(M > 0 && E > 0 && E % 32 == 0 && E <= 256 * 8);
    const int blocks = ln_fwd_blocks(M, (E + 255) / 256 <= 5 ? 2 : 1);
#define VP_LNP(CH)                                                                                                   \
    hipLaunchKernelGGL((ln_fwd_kernel<CH, true, (CH <= 5 ? 2 : 1)>), dim3(blocks), dim3(256), 0, stream, x, gamma, beta, \
                       (bf16*)y_pair, y_f32, (float*)nullptr, (float*)nullptr, M, E, eps, relu)
    switch ((E + 255) / 256) {
        case 1: VP_LNP(1); break;
        case 2: VP_LNP(2); break;
        case 3: VP_LNP(3); break;
        case 4: VP_LNP(4); break;
        case 5: VP_LNP(5); break;
        case 6: VP_LNP(6); break;
        case 7: VP_LNP(7); break;
        default: VP_LNP(8); break;
    }
#undef VP_LNP
    VP_LAUNCH_RET();
}

extern "C" int vp_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                                const float* gamma, const float* dres, float* dx, void* dxb_bf16,
                                const float* rowscale, int rows_per_scale, float* dgamma, float* dbeta,
                                float* dcolsum, int M, int E, float* workspace, size_t workspace_floats,
                                hipStream_t stream) {
    VP_CHECK_ARG(dy_bf16 && x && mean && rstd && gamma && (dx || dxb_bf16) && workspace);
    VP_CHECK_ARG(M > 0 && E > 0 && E % 4 == 0 && E <= 256 * 8);
    if (rowscale) VP_CHECK_ARG(rows_per_scale > 0);
    const int blocks = ln_bwd_blocks(M, E, workspace_floats);
    VP_CHECK_ARG(blocks >= 1);
#define VP_LNB(CH)                                                                                                \
    hipLaunchKernelGGL(ln_bwd_kernel<CH>, dim3(blocks), dim3(512), 0, stream, (const bf16*)dy_bf16, x, mean, rstd, \
                       gamma, dres, dx, (bf16*)dxb_bf16, rowscale, rows_per_scale, workspace, M, E)
    switch ((E + 255) / 256) {
        case 1: VP_LNB(1); break;
        case 2: VP_LNB(2); break;
        case 3: VP_LNB(3); break;
        case 4: VP_LNB(4); break;
        case 5: VP_LNB(5); break;
        case 6: VP_LNB(6); break;
        case 7: VP_LNB(7); break;
        default: VP_LNB(8); break;
    }
#undef VP_LNB
    // all three targets NULL: the partial sums stay in `workspace` for a later vp_layernorm_bwd_finalize
    if (dgamma || dbeta || dcolsum)
        hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((3 * E + 15) / 16), dim3(256), 0, stream, workspace, blocks, E,
                           dgamma, dbeta, dcolsum);
    VP_LAUNCH_RET();
}

extern "C" int vp_layernorm_bwd_static(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                                       const float* gamma, const float* dres, float* dx, void* dxb_bf16,
                                       const float* rowscale, int rows_per_scale, float* dgamma, float* dbeta,
                                       float* dcolsum, int M, int E, hipStream_t stream) {
    VP_CHECK_ARG(E > 0 && M > 0);
    if (E > VP_LN_STATIC_MAX_E) return VP_ERR_UNSUPPORTED;      // the static scratch holds rows of <= 1280 channels
    VP_CHECK_ARG(dgamma || dbeta || dcolsum);                   // no deferred sums: the scratch is shared by every call
    float* ws = nullptr;
    const int rc = static_scratch(&ws);
    if (rc != VP_OK) return rc;
    // every workgroup overwrites its own [3][E] slice and the finalize pass reads the slices of this launch
    // only, so nothing of an earlier call has to be cleared
    return vp_layernorm_bwd(dy_bf16, x, mean, rstd, gamma, dres, dx, dxb_bf16, rowscale, rows_per_scale, dgamma, dbeta,
                            dcolsum, M, E, ws, (size_t)VP_LN_STATIC_MAX_BLOCKS * 3 * VP_LN_STATIC_MAX_E, stream);
}

extern "C" int vp_ln_fold_weights(const float* W, const float* gamma, const float* beta, const float* bias, void* Wg,
                                  float* s, float* c, int N, int K, hipStream_t stream) {
    VP_CHECK_ARG(W && gamma && beta && Wg && s && c && N > 0 && K > 0);
    hipLaunchKernelGGL(ln_fold_weights_kernel, dim3(N), dim3(256), 0, stream, W, gamma, beta, bias, (bf16*)Wg, s, c, K);
    VP_LAUNCH_RET();
}
