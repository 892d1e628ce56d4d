// BatchNorm (channels-last), heatmap MSE loss, arg-max for PCK, gradient norm and the
// fused AdamW step.  All HBM-bound streaming/reduction kernels.
#include "common.h"
#include "vitpose_hip.h"

namespace {

inline int grid_for(size_t work_items, int cap = 4096) {
    size_t g = (work_items + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g ? g : 1));
}

// ---------------- BatchNorm over rows of a channels-last bf16 [M][C] matrix ----------------
// per-channel sum and sum of squares (f32 atomics into sums[0:C] and sums[C:2C])
__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16* __restrict__ z, float* __restrict__ sums, int M,
                                                       int C, int rows_per_block) {
    __shared__ float red[2][8][256 + 8];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int col = (blockIdx.x * 32 + cx) * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < C) {
        const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
#pragma unroll 4
        for (int r = r0 + ry; r < r1; r += 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(z + (size_t)r * C + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = (float)v[j];
                s[j] += f;
                q[j] += f * f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[0][ry][cx * 8 + j] = s[j];
        red[1][ry][cx * 8 + j] = q[j];
    }
    __syncthreads();
    const int c = threadIdx.x, gc = blockIdx.x * 256 + c;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        a += red[0][r][c];
        b += red[1][r][c];
    }
    if (gc < C) {
        atomicAdd(sums + gc, a);
        atomicAdd(sums + C + gc, b);
    }
}

// training: batch mean / biased var -> scale, shift, saved mean / invstd; running stats updated
// with momentum and the unbiased variance (torch.nn.BatchNorm2d).  eval (sums == NULL): scale /
// shift from the running statistics.
__global__ void bn_finalize_kernel(const float* __restrict__ sums, int M, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float momentum, float eps,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (sums) {
        mean = sums[c] / (float)M;
        var = fmaxf(sums[C + c] / (float)M - mean * mean, 0.f);
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * ((float)M / (float)(M > 1 ? M - 1 : 1));
        }
    } else {
        mean = running_mean[c];
        var = running_var[c];
    }
    const float inv = rsqrtf(var + eps);
    const float sc = gamma[c] * inv;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (save_mean) save_mean[c] = mean;
    if (save_invstd) save_invstd[c] = inv;
}

__global__ __launch_bounds__(256) void bn_apply_relu_kernel(const bf16* __restrict__ z, bf16* __restrict__ a,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, size_t n8, int C) {
    // a thread's 8 channels only change when the grid stride is not a multiple of C (never for
    // C | 2048): the per-channel coefficients stay in registers across iterations
    int cc = -1;
    float sc[8], sh[8];
    // the 8-channel chunk of element i within its row, kept incrementally (a 64-bit modulo per iteration
    // costs more VALU time than the iteration's memory traffic takes)
    const unsigned cpr = (unsigned)C / 8, stride = gridDim.x * blockDim.x, cstep = stride % cpr;
    unsigned ci = (blockIdx.x * blockDim.x + threadIdx.x) % cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride, ci += cstep) {
        if (ci >= cpr) ci -= cpr;
        const int c = (int)ci * 8;
        if (c != cc) {
            cc = c;
#pragma unroll
            for (int j = 0; j < 8; ++j) { sc[j] = scale[c + j]; sh[j] = shift[c + j]; }
        }
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(z + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaf((float)v[j], sc[j], sh[j]), 0.f);
        *reinterpret_cast<bf16x8*>(a + i * 8) = o;
    }
}

// backward reduce: g = grad wrt the BN output (ReLU mask already applied);
// dbeta[c] += sum g, dgamma[c] += sum g * xhat, xhat = (z - mean) * invstd
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16* __restrict__ g, const bf16* __restrict__ z,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int M, int C, int rows_per_block) {
    __shared__ float red[2][8][256 + 8];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int col = (blockIdx.x * 32 + cx) * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < C) {
        float mu[8], is[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mu[j] = mean[col + j];
            is[j] = invstd[col + j];
        }
        const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
#pragma unroll 4
        for (int r = r0 + ry; r < r1; r += 8) {
            const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + (size_t)r * C + col);
            const bf16x8 zv = *reinterpret_cast<const bf16x8*>(z + (size_t)r * C + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = (float)gv[j];
                s[j] += f;
                q[j] += f * ((float)zv[j] - mu[j]) * is[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[0][ry][cx * 8 + j] = s[j];
        red[1][ry][cx * 8 + j] = q[j];
    }
    __syncthreads();
    const int c = threadIdx.x, gc = blockIdx.x * 256 + c;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        a += red[0][r][c];
        b += red[1][r][c];
    }
    if (gc < C) {
        atomicAdd(dbeta + gc, a);
        atomicAdd(dgamma + gc, b);
    }
}

// dz = gamma * invstd * (g - dbeta/M - xhat * dgamma/M) = k1 * g - k2 - k3 * z  per channel
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16* __restrict__ g, const bf16* __restrict__ z,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, bf16* __restrict__ dz,
                                                           size_t n8, int C, float invM) {
    int cc = -1;
    float k1[8], k2[8], k3[8];
    const unsigned cpr = (unsigned)C / 8, stride = gridDim.x * blockDim.x, cstep = stride % cpr;
    unsigned ci = (blockIdx.x * blockDim.x + threadIdx.x) % cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride, ci += cstep) {
        if (ci >= cpr) ci -= cpr;
        const int c = (int)ci * 8;
        if (c != cc) {
            cc = c;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float is = invstd[c + j], gi = gamma[c + j] * is;
                const float dgm = dgamma[c + j] * invM * is;     // coefficient of (z - mean)
                k1[j] = gi;
                k3[j] = gi * dgm;
                k2[j] = gi * (dbeta[c + j] * invM - dgm * mean[c + j]);
            }
        }
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + i * 8);
        const bf16x8 zv = *reinterpret_cast<const bf16x8*>(z + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)(k1[j] * (float)gv[j] - k2[j] - k3[j] * (float)zv[j]);
        *reinterpret_cast<bf16x8*>(dz + i * 8) = o;
    }
}

// ---------------- heatmap loss (JointsMSELoss) ----------------
// One thread per (image, heat-map pixel).  Reads the K predictions of its pixel from the
// channels-last GEMM output [B*HW][ldh] f32, optionally writes them as NCHW f32, and, when a
// target is given, accumulates  sum_k (w_bk (p - t))^2 * coef  into *loss and writes
// dL/dp = 2 w_bk^2 (p - t) coef as channels-last bf16 [B*HW][ldh] (pad columns zeroed).
__global__ __launch_bounds__(256) void heatmap_loss_kernel(const float* __restrict__ heat, int ldh,
                                                           float* __restrict__ heat_nchw,
                                                           const float* __restrict__ target,
                                                           const float* __restrict__ tw, float* __restrict__ loss,
                                                           bf16* __restrict__ dheat, int B, int K, int HW,
                                                           float coef) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float part = 0.f;
    if (i < (size_t)B * HW) {
        const int b = (int)(i / HW), hw = (int)(i % HW);
        const float* hp = heat + i * ldh;
        bf16* dp = dheat ? dheat + i * ldh : nullptr;
        // one pixel per thread, 8 channels at a time: two 16-byte loads, one 16-byte gradient store
        // (per-channel NCHW accesses are contiguous across the wave's 64 pixels)
        for (int k0 = 0; k0 < ldh; k0 += 8) {
            bf16x8 g8;
#pragma unroll
            for (int j = 0; j < 8; ++j) g8[j] = (bf16)0.f;
            if (k0 < K) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(hp + k0);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(hp + k0 + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = k0 + j;
                    if (k < K) {
                        const float p = j < 4 ? lo[j] : hi[j - 4];
                        const size_t o = ((size_t)b * K + k) * HW + hw;
                        if (heat_nchw) heat_nchw[o] = p;
                        if (target) {
                            const float w = tw ? tw[b * K + k] : 1.f;
                            const float d = w * (p - target[o]);
                            part += d * d;
                            g8[j] = (bf16)(2.f * w * d * coef);
                        }
                    }
                }
            }
            if (dp) *reinterpret_cast<bf16x8*>(dp + k0) = g8;
        }
    }
    if (target && loss) {
        part = wave_sum(part);
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * coef);
    }
}

// ---------------- per-row arg-max (first maximum, like numpy.argmax) ----------------
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int L, int* __restrict__ idx,
                                                          float* __restrict__ val) {
    const float* row = x + (size_t)blockIdx.x * L;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < L; i += 256) {
        const float v = row[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    __shared__ float sv[256];
    __shared__ int si[256];
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float v = sv[threadIdx.x + s];
            const int j = si[threadIdx.x + s];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && j < si[threadIdx.x])) {
                sv[threadIdx.x] = v;
                si[threadIdx.x] = j;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        idx[blockIdx.x] = si[0];
        if (val) val[blockIdx.x] = sv[0];
    }
}

// ---------------- gradient norm + AdamW over flat buffers ----------------
// Deterministic variant: workgroup b writes its partial sum to partial[b]; sumsq_final_kernel adds the partial sums
// in a fixed order.  Data-parallel replicas compute the clip factor from identical (all-reduced) gradients and must
// arrive at identical bits, or their weights drift apart step by step (float atomics add in arrival order).
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, size_t n4,
                                                            float* __restrict__ partial) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// the same over a chunk table (the optimizer's: {start, length} in elements, lengths multiples of 4): workgroup b
// takes chunks b, b + gridDim.x, ... -- a fixed assignment, so the sum has a fixed order
__global__ __launch_bounds__(256) void sumsq_chunks_partial_kernel(const float* __restrict__ x,
                                                                   const long* __restrict__ chunk_start,
                                                                   const int* __restrict__ chunk_len, int nchunks,
                                                                   float* __restrict__ partial) {
    // two loads in flight per lane and pass (independent partial sums, combined in a fixed order)
    float s0 = 0.f, s1 = 0.f;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const float* p = x + chunk_start[c];
        const int len = chunk_len[c];
        for (int i = threadIdx.x * 4; i < len; i += 2 * 256 * 4) {
            const int i2 = i + 256 * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + i);
            const f32x4 w = i2 < len ? *reinterpret_cast<const f32x4*>(p + i2) : f32x4{0.f, 0.f, 0.f, 0.f};
            s0 += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            s1 += w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3];
        }
    }
    float s = wave_sum(s0 + s1);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_assign_kernel(const float* __restrict__ partial, int n,
                                                           float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

// Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): counter-based, so element i of
// step `ctr` is a pure function of (seed, ctr, i)
__device__ __forceinline__ uint32_t philox_first(uint64_t seed, uint64_t ctr, uint32_t idx) {
    uint32_t c0 = idx, c1 = 0, c2 = (uint32_t)ctr, c3 = (uint32_t)(ctr >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0;
}

// DropPath scales of one step: dp[i][j][b] = floor(keep_i + U) / keep_i, keep_i = 1 - rates[i]; rate 0 (Block 0, a
// Block in eval mode) gives exactly 1
__global__ __launch_bounds__(256) void drop_path_kernel(float* __restrict__ dp, const float* __restrict__ rates,
                                                        int per_block, int total, uint64_t seed, uint64_t ctr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float keep = 1.f - rates[i / per_block];
    const float u = (float)(philox_first(seed, ctr, (uint32_t)i) >> 8) * (1.0f / 16777216.0f);   // [0, 1), 24 bits
    dp[i] = keep >= 1.f ? 1.f : floorf(keep + u) / keep;
}

// several small accumulators zeroed by ONE launch (a memset node per region costs ~5 us of launch each)
struct ZeroRegions {
    float* p[VP_ZERO_MAX_REGIONS];
    int n[VP_ZERO_MAX_REGIONS];       // floats
    int count;
};
__global__ __launch_bounds__(256) void zero_regions_kernel(ZeroRegions r) {
    float* p = r.p[blockIdx.y];
    const int n = r.n[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.f;
}

// out[t][e] = a[1 + t][e] + a[0][e]: the position table every token row receives (vit.py:320)
__global__ __launch_bounds__(256) void pos_combine_kernel(const float* __restrict__ pe, float* __restrict__ out, int T,
                                                          int E) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T * E) out[i] = pe[E + i] + pe[i % E];
}

__global__ void add_i64_kernel(long long** p, int count, long long v) {
    if ((int)threadIdx.x < count) *p[threadIdx.x] += v;
}

__global__ __launch_bounds__(256) void scale_bf16_kernel(bf16* __restrict__ x, size_t n8, const float* __restrict__ scale) {
    const float s = *scale;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        bf16x8 v = *reinterpret_cast<bf16x8*>(x + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] * s);
        *reinterpret_cast<bf16x8*>(x + i * 8) = v;
    }
}

__global__ __launch_bounds__(256) void scale_f32_kernel(float* __restrict__ x, size_t n4, const float* __restrict__ scale) {
    const float s = *scale;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = *reinterpret_cast<f32x4*>(x + i * 4);
        *reinterpret_cast<f32x4*>(x + i * 4) = v * s;
    }
}

// out = bf16(sum of up to 8 bf16 arrays, added in f32): several heads on one feature map (TopDownMoE) hand the
// backbone the sum of their feature gradients
struct SumSrc {
    const bf16* p[8];
};
__global__ __launch_bounds__(256) void sum_bf16_kernel(SumSrc src, int count, bf16* __restrict__ out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int k = 0; k < count; ++k) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src.p[k] + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
        *reinterpret_cast<bf16x8*>(out + i * 8) = o;
    }
}

// out[b][:] = x[b][:] if source[b] == which else 0 (row length n): the per-head selection of targets / target weights
__global__ __launch_bounds__(256) void mask_rows_kernel(const float* __restrict__ x, const int* __restrict__ source,
                                                        int which, float* __restrict__ out, int B, size_t n) {
    const size_t total = (size_t)B * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        out[i] = source[i / n] == which ? x[i] : 0.f;
}

// Backward of the identity that stands where the final LayerNorm would be (ViT(last_norm=False)): the feature gradient
// is the residual-stream gradient (f32 copy) and, scaled by the last Block's DropPath row scale, the bf16 operand of that
// Block's fc2 data / weight gradients.
__global__ __launch_bounds__(256) void identity_norm_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ rowscale,
                                                                int rows_per_scale, float* __restrict__ dx,
                                                                bf16* __restrict__ dxb, int M, int E8) {
    const size_t n8 = (size_t)M * E8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / E8);
        const float s = rowscale ? rowscale[row / rows_per_scale] : 1.f;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + i * 8);
        f32x4 lo, hi;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[j] = (float)v[j];
            hi[j] = (float)v[4 + j];
            o[j] = (bf16)(lo[j] * s);
            o[4 + j] = (bf16)(hi[j] * s);
        }
        *reinterpret_cast<f32x4*>(dx + i * 8) = lo;
        *reinterpret_cast<f32x4*>(dx + i * 8 + 4) = hi;
        *reinterpret_cast<bf16x8*>(dxb + i * 8) = o;
    }
}

// row 0 of a [1 + T][E] position-embedding gradient: the sum of rows 1 .. T (vit.py:320 adds pos_embed[:, :1]
// to every token)
// (one workgroup per 64 columns: 16 row groups x 16 four-column lanes, summed through LDS in a fixed order)
__global__ __launch_bounds__(256) void pos_row0_kernel(float* __restrict__ gpos, int T, int E) {
    __shared__ f32x4 red[16][16];
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int e = blockIdx.x * 64 + cq * 4;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (e < E) {
#pragma unroll 4
        for (int t = 1 + rg; t <= T; t += 16) s += *reinterpret_cast<const f32x4*>(gpos + (size_t)t * E + e);
    }
    red[rg][cq] = s;
    __syncthreads();
    if (rg == 0 && e < E) {
        f32x4 t = red[0][cq];
#pragma unroll
        for (int r = 1; r < 16; ++r) t += red[r][cq];
        *reinterpret_cast<f32x4*>(gpos + e) = t;
    }
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int n,
                                                          float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out += red[0];
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n4, float* __restrict__ out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// One block per chunk of one parameter tensor; chunk table rows = {start, length} (elements,
// multiples of 4), with a per-chunk lr scale and weight decay.  torch.optim.AdamW update order.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16* __restrict__ p_bf16, const long* __restrict__ chunk_start,
                                                    const int* __restrict__ chunk_len,
                                                    const float* __restrict__ chunk_lr_scale,
                                                    const float* __restrict__ chunk_wd, float lr, float beta1,
                                                    float beta2, float eps, float bc1, float bc2_sqrt,
                                                    const float* __restrict__ gnorm_sq, float max_norm,
                                                    float grad_scale) {
    const long s0 = chunk_start[blockIdx.x];
    const int len = chunk_len[blockIdx.x];
    const float lr_c = lr * chunk_lr_scale[blockIdx.x], wd = chunk_wd[blockIdx.x];
    float clip = grad_scale;
    if (gnorm_sq && max_norm > 0.f) {
        const float tn = sqrtf(*gnorm_sq) * grad_scale;
        clip *= fminf(1.f, max_norm / (tn + 1e-6f));
    }
    const float step = lr_c / bc1;
    auto update = [&](long o, f32x4 pv, f32x4 gv, f32x4 mv, f32x4 vv) {
        gv *= clip;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pv[r] *= 1.f - lr_c * wd;
            mv[r] += (1.f - beta1) * (gv[r] - mv[r]);
            vv[r] = beta2 * vv[r] + (1.f - beta2) * gv[r] * gv[r];
            pv[r] -= step * mv[r] / (sqrtf(vv[r]) / bc2_sqrt + eps);
        }
        *reinterpret_cast<f32x4*>(p + o) = pv;
        *reinterpret_cast<f32x4*>(m + o) = mv;
        *reinterpret_cast<f32x4*>(v + o) = vv;
        if (p_bf16) *reinterpret_cast<bf16x4*>(p_bf16 + o) = pack4(pv[0], pv[1], pv[2], pv[3]);
    };
    // two float4 per lane and pass: all eight loads are issued before the first store
    for (int i = threadIdx.x * 4; i < len; i += 2 * 256 * 4) {
        const long o = s0 + i, o2 = o + 256 * 4;
        const bool two = i + 256 * 4 < len;
        const f32x4 pa = *reinterpret_cast<const f32x4*>(p + o), ga = *reinterpret_cast<const f32x4*>(g + o);
        const f32x4 ma = *reinterpret_cast<const f32x4*>(m + o), va = *reinterpret_cast<const f32x4*>(v + o);
        f32x4 pb = pa, gb = ga, mb = ma, vb = va;
        if (two) {
            pb = *reinterpret_cast<const f32x4*>(p + o2);
            gb = *reinterpret_cast<const f32x4*>(g + o2);
            mb = *reinterpret_cast<const f32x4*>(m + o2);
            vb = *reinterpret_cast<const f32x4*>(v + o2);
        }
        update(o, pa, ga, ma, va);
        if (two) update(o2, pb, gb, mb, vb);
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        *reinterpret_cast<bf16x4*>(y + i * 4) = pack4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace

extern "C" int vp_bn_stats(const void* z, float* sums, int M, int C, hipStream_t stream) {
    VP_CHECK_ARG(z && sums && M > 0 && C > 0 && C % 8 == 0);
    const int gx = (C + 255) / 256;
    int gy = (M + 255) / 256;
    if (gy > 512) gy = 512;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(gx, gy), dim3(256), 0, stream, (const bf16*)z, sums, M, C, (M + gy - 1) / gy);
    VP_LAUNCH_RET();
}

extern "C" int vp_bn_finalize(const float* sums, int M, int C, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* scale,
                              float* shift, float* save_mean, float* save_invstd, hipStream_t stream) {
    VP_CHECK_ARG(C > 0 && gamma && beta && scale && shift && (sums || (running_mean && running_var)));
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sums, M, C, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd);
    VP_LAUNCH_RET();
}

extern "C" int vp_bn_apply_relu(const void* z, void* a, const float* scale, const float* shift, int M, int C,
                                hipStream_t stream) {
    VP_CHECK_ARG(z && a && scale && shift && M > 0 && C % 8 == 0);
    const size_t n8 = (size_t)M * C / 8;
    hipLaunchKernelGGL(bn_apply_relu_kernel, dim3(grid_for(n8)), dim3(256), 0, stream, (const bf16*)z, (bf16*)a, scale,
                       shift, n8, C);
    VP_LAUNCH_RET();
}

static int bn_bwd_launch(const void* g, const void* z, const float* mean, const float* invstd, const float* gamma,
                         float* dgamma, float* dbeta, void* dz, int M, int C, bool batch_stats, hipStream_t stream) {
    VP_CHECK_ARG(g && z && mean && invstd && gamma && dgamma && dbeta && dz && M > 0 && C % 8 == 0);
    const int gx = (C + 255) / 256;
    int gy = (M + 255) / 256;
    if (gy > 512) gy = 512;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(gx, gy), dim3(256), 0, stream, (const bf16*)g, (const bf16*)z, mean,
                       invstd, dgamma, dbeta, M, C, (M + gy - 1) / gy);
    const size_t n8 = (size_t)M * C / 8;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(n8)), dim3(256), 0, stream, (const bf16*)g, (const bf16*)z,
                       mean, invstd, gamma, dgamma, dbeta, (bf16*)dz, n8, C, batch_stats ? 1.f / (float)M : 0.f);
    VP_LAUNCH_RET();
}

extern "C" int vp_bn_bwd(const void* g, const void* z, const float* mean, const float* invstd, const float* gamma,
                         float* dgamma, float* dbeta, void* dz, int M, int C, hipStream_t stream) {
    return bn_bwd_launch(g, z, mean, invstd, gamma, dgamma, dbeta, dz, M, C, true, stream);
}

// Statistics that do not depend on the batch (eval mode: running mean / variance): the mean- and variance-gradient
// terms of the batch-statistics formula vanish, dz = g * gamma * invstd; d gamma and d beta are the same sums.
extern "C" int vp_bn_bwd_eval(const void* g, const void* z, const float* mean, const float* invstd, const float* gamma,
                              float* dgamma, float* dbeta, void* dz, int M, int C, hipStream_t stream) {
    return bn_bwd_launch(g, z, mean, invstd, gamma, dgamma, dbeta, dz, M, C, false, stream);
}

// The same for K <= 32 maps in rows of at most 64 padded columns (every ViTPose body model), with all three arrays
// moved in whole cache lines: a workgroup takes 256 consecutive pixels, reads the first ceil(K / 4) 16-byte chunks of
// their channels-last rows chunk by chunk into LDS, then works one pixel per thread on the NCHW side (consecutive
// threads = consecutive pixels of one map: coalesced), and writes the padded bf16 gradient rows back as one
// contiguous 32 KiB piece.  (One pixel per thread on the channels-last side touches 64 cache lines per instruction.)
constexpr int HL_PIX = 256, HL_MAXQ = 8, HL_GP = 72;    // pixels per workgroup, 16-byte chunks per row, bf16 pitch of the gradient tile
__global__ __launch_bounds__(256) void heatmap_loss_tiled_kernel(const float* __restrict__ heat, int ldh,
                                                                 float* __restrict__ heat_nchw,
                                                                 const float* __restrict__ target,
                                                                 const float* __restrict__ tw, float* __restrict__ loss,
                                                                 bf16* __restrict__ dheat, int B, int K, int HW,
                                                                 float coef) {
    __shared__ float tile[HL_PIX * (4 * HL_MAXQ + 1)];
    __shared__ __attribute__((aligned(16))) bf16 gt[HL_PIX * HL_GP];
    const size_t p0 = (size_t)blockIdx.x * HL_PIX, npix = (size_t)B * HW;
    const int nq = (K + 3) >> 2, pitch = 4 * nq + 1;
    for (int i = threadIdx.x; i < HL_PIX * nq; i += 256) {
        const int px = i / nq, q = i - px * nq;
        if (p0 + px < npix) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(heat + (p0 + px) * ldh + 4 * q);
            float* d = tile + px * pitch + 4 * q;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
    }
    if (dheat) {   // padded columns of the gradient rows are zero
        for (int i = threadIdx.x; i < HL_PIX * HL_GP / 8; i += 256) {
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
            *reinterpret_cast<bf16x8*>(gt + i * 8) = z;
        }
    }
    __syncthreads();
    float part = 0.f;
    const size_t i = p0 + threadIdx.x;
    if (i < npix) {
        const int b = (int)(i / HW), hw = (int)(i % HW);
        const float* row = tile + threadIdx.x * pitch;
        for (int k = 0; k < K; ++k) {
            const float p = row[k];
            const size_t o = ((size_t)b * K + k) * HW + hw;
            if (heat_nchw) heat_nchw[o] = p;
            if (target) {
                const float w = tw ? tw[b * K + k] : 1.f;
                const float d = w * (p - target[o]);
                part += d * d;
                gt[threadIdx.x * HL_GP + k] = (bf16)(2.f * w * d * coef);
            }
        }
    }
    if (dheat) {
        __syncthreads();
        const int cpr = ldh >> 3;           // 16-byte chunks per gradient row
        for (int j = threadIdx.x; j < HL_PIX * cpr; j += 256) {
            const int px = j / cpr, c = j - px * cpr;
            if (p0 + px < npix)
                *reinterpret_cast<uint4*>(dheat + (p0 + px) * ldh + 8 * c) = *reinterpret_cast<const uint4*>(gt + px * HL_GP + 8 * c);
        }
    }
    if (target && loss) {
        part = wave_sum(part);
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * coef);
    }
}

extern "C" int vp_heatmap_loss(const float* heat, int ldh, float* heat_nchw, const float* target, const float* tw,
                               float* loss, void* dheat, int B, int K, int HW, float loss_weight, hipStream_t stream) {
    VP_CHECK_ARG(heat && B > 0 && K > 0 && K <= ldh && HW > 0 && ldh % 8 == 0);
    VP_CHECK_ARG(((uintptr_t)heat & 15) == 0 && ((uintptr_t)dheat & 15) == 0);
    VP_CHECK_ARG(heat_nchw || target);
    const float coef = loss_weight / ((float)B * (float)K * (float)HW);
    const size_t n = (size_t)B * HW;
    if (K <= 4 * HL_MAXQ && ldh <= 64) {
        hipLaunchKernelGGL(heatmap_loss_tiled_kernel, dim3((unsigned)((n + HL_PIX - 1) / HL_PIX)), dim3(256), 0, stream,
                           heat, ldh, heat_nchw, target, tw, loss, (bf16*)dheat, B, K, HW, coef);
        VP_LAUNCH_RET();
    }
    hipLaunchKernelGGL(heatmap_loss_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, heat, ldh,
                       heat_nchw, target, tw, loss, (bf16*)dheat, B, K, HW, coef);
    VP_LAUNCH_RET();
}

extern "C" int vp_argmax_rows(const float* x, int rows, int L, int* idx, float* val, hipStream_t stream) {
    VP_CHECK_ARG(x && idx && rows > 0 && L > 0);
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, stream, x, L, idx, val);
    VP_LAUNCH_RET();
}

extern "C" int vp_sumsq_f32_det(const float* x, size_t n, float* out, float* partials, int npartials,
                                hipStream_t stream) {
    VP_CHECK_ARG(x && out && partials && n > 0 && n % 4 == 0 && npartials >= 1);
    const int blocks = npartials < 1024 ? npartials : 1024;      // fixed for a given (n, npartials): same order every call
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, stream, x, n / 4, partials);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, partials, blocks, out);
    VP_LAUNCH_RET();
}

extern "C" int vp_sumsq_f32(const float* x, size_t n, float* out, hipStream_t stream) {
    VP_CHECK_ARG(x && out && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 4, 1024)), dim3(256), 0, stream, x, n / 4, out);
    VP_LAUNCH_RET();
}

extern "C" int vp_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, const long* chunk_start,
                             const int* chunk_len, const float* chunk_lr_scale, const float* chunk_wd, int nchunks,
                             float lr, float beta1, float beta2, float eps, int step, const float* gnorm_sq,
                             float max_norm, float grad_scale, hipStream_t stream) {
    VP_CHECK_ARG(p && g && m && v && chunk_start && chunk_len && chunk_lr_scale && chunk_wd && nchunks > 0 && step > 0);
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, stream, p, g, m, v, (bf16*)p_bf16, chunk_start,
                       chunk_len, chunk_lr_scale, chunk_wd, lr, beta1, beta2, eps, bc1, bc2s, gnorm_sq, max_norm,
                       grad_scale);
    VP_LAUNCH_RET();
}

extern "C" int vp_cast_f32_bf16(const float* x, void* y, size_t n, hipStream_t stream) {
    VP_CHECK_ARG(x && y && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, x, (bf16*)y, n / 4);
    VP_LAUNCH_RET();
}

extern "C" int vp_sumsq_chunks_det(const float* x, const long* chunk_start, const int* chunk_len, int nchunks, float* out,
                                   float* partials, int npartials, hipStream_t stream) {
    VP_CHECK_ARG(x && chunk_start && chunk_len && nchunks > 0 && out && partials && npartials >= 1);
    int blocks = npartials < 2048 ? npartials : 2048;      // eight 256-thread workgroups per CU
    if (blocks > nchunks) blocks = nchunks;
    hipLaunchKernelGGL(sumsq_chunks_partial_kernel, dim3(blocks), dim3(256), 0, stream, x, chunk_start, chunk_len, nchunks,
                       partials);
    hipLaunchKernelGGL(sumsq_assign_kernel, dim3(1), dim3(256), 0, stream, partials, blocks, out);
    VP_LAUNCH_RET();
}

extern "C" int vp_drop_path_scales(float* dp, const float* rates, int depth, int per_block, unsigned long long seed,
                                   unsigned long long counter, hipStream_t stream) {
    VP_CHECK_ARG(dp && rates && depth > 0 && per_block > 0);
    const int total = depth * per_block;
    hipLaunchKernelGGL(drop_path_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, dp, rates, per_block, total,
                       (uint64_t)seed, (uint64_t)counter);
    VP_LAUNCH_RET();
}

extern "C" int vp_scale_bf16(void* x, size_t n, const float* scale_dev, hipStream_t stream) {
    VP_CHECK_ARG(x && scale_dev && n > 0 && n % 8 == 0 && ((uintptr_t)x & 15) == 0);
    hipLaunchKernelGGL(scale_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (bf16*)x, n / 8, scale_dev);
    VP_LAUNCH_RET();
}

extern "C" int vp_identity_norm_bwd(const void* dy, const float* rowscale, int rows_per_scale, float* dx, void* dxb,
                                    int M, int E, hipStream_t stream) {
    VP_CHECK_ARG(dy && dx && dxb && M > 0 && E > 0 && E % 8 == 0 && (!rowscale || rows_per_scale > 0));
    VP_CHECK_ARG((((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)dxb) & 15) == 0);
    hipLaunchKernelGGL(identity_norm_bwd_kernel, dim3(grid_for((size_t)M * (E / 8))), dim3(256), 0, stream,
                       (const bf16*)dy, rowscale, rows_per_scale, dx, (bf16*)dxb, M, E / 8);
    VP_LAUNCH_RET();
}

extern "C" int vp_scale_f32(float* x, size_t n, const float* scale_dev, hipStream_t stream) {
    VP_CHECK_ARG(x && scale_dev && n > 0 && n % 4 == 0 && ((uintptr_t)x & 15) == 0);
    hipLaunchKernelGGL(scale_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, x, n / 4, scale_dev);
    VP_LAUNCH_RET();
}

extern "C" int vp_sum_bf16(const void* const* srcs_host, int count, void* out, size_t n, hipStream_t stream) {
    VP_CHECK_ARG(srcs_host && out && count > 0 && count <= 8 && n > 0 && n % 8 == 0 && ((uintptr_t)out & 15) == 0);
    SumSrc s{};
    for (int i = 0; i < count; ++i) {
        VP_CHECK_ARG(srcs_host[i] && ((uintptr_t)srcs_host[i] & 15) == 0);
        s.p[i] = (const bf16*)srcs_host[i];
    }
    hipLaunchKernelGGL(sum_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, s, count, (bf16*)out, n / 8);
    VP_LAUNCH_RET();
}

extern "C" int vp_mask_rows_f32(const float* x, const int* source, int which, float* out, int B, size_t n,
                                hipStream_t stream) {
    VP_CHECK_ARG(x && source && out && B > 0 && n > 0);
    hipLaunchKernelGGL(mask_rows_kernel, dim3(grid_for((size_t)B * n)), dim3(256), 0, stream, x, source, which, out, B, n);
    VP_LAUNCH_RET();
}

extern "C" int vp_zero(void* p, size_t bytes, hipStream_t stream) {
    VP_CHECK_ARG(p && bytes > 0);
    const hipError_t e = hipMemsetAsync(p, 0, bytes, stream);
    return e == hipSuccess ? VP_OK : (int)e;
}

extern "C" int vp_pos_embed_grad(const float* dx, float* gpos, int B, int T, int E, hipStream_t stream) {
    VP_CHECK_ARG(dx && gpos && B > 0 && T > 0 && E > 0 && E % 4 == 0);
    const int rc = vp_batch_sum_f32(dx, gpos + E, B, (size_t)T * E, stream);      // rows 1 .. T
    if (rc != VP_OK) return rc;
    hipLaunchKernelGGL(pos_row0_kernel, dim3((E + 63) / 64), dim3(256), 0, stream, gpos, T, E);
    VP_LAUNCH_RET();
}

extern "C" int vp_zero_regions(float* const* ptrs_host, const int* floats_host, int count, hipStream_t stream) {
    VP_CHECK_ARG(ptrs_host && floats_host && count > 0 && count <= VP_ZERO_MAX_REGIONS);
    ZeroRegions r{};
    int mx = 0;
    for (int i = 0; i < count; ++i) {
        VP_CHECK_ARG(ptrs_host[i] && floats_host[i] > 0);
        r.p[i] = ptrs_host[i];
        r.n[i] = floats_host[i];
        mx = floats_host[i] > mx ? floats_host[i] : mx;
    }
    r.count = count;
    int gx = (mx + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(zero_regions_kernel, dim3(gx, count), dim3(256), 0, stream, r);
    VP_LAUNCH_RET();
}

extern "C" int vp_pos_combine(const float* pos_embed, float* out, int T, int E, hipStream_t stream) {
    VP_CHECK_ARG(pos_embed && out && T > 0 && E > 0);
    hipLaunchKernelGGL(pos_combine_kernel, dim3((T * E + 255) / 256), dim3(256), 0, stream, pos_embed, out, T, E);
    VP_LAUNCH_RET();
}
