// LayerNorm forward / backward over the channel dimension of token-major activations.
// One wavefront per row: the row lives in registers (float4 chunks strided by 64 lanes),
// statistics are fp32 wave reductions, every global access is 8 or 16 bytes per lane.
// HBM-bound: forward moves 4E (x) + 2E (y) bytes per row; backward 4E+2E in, 4E+2E out.
#include "common.h"
#include "vitpose_hip.h"

namespace {

constexpr int MAXC = 8;  // float4 chunks per lane -> E <= 2048

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16* __restrict__ y,
                                                     float* __restrict__ y32, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int M, int E, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = E >> 2;
    const float* xr = x + (size_t)row * E;
    f32x4 v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c);
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = v[i][r] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)E + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * c);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - mean) * rstd * g[r] + b[r];
            if (y) *reinterpret_cast<bf16x4*>(y + (size_t)row * E + 4 * c) = pack4(o[0], o[1], o[2], o[3]);
            if (y32) *reinterpret_cast<f32x4*>(y32 + (size_t)row * E + 4 * c) = o;
        }
    }
}

// Backward.  Each wave walks rows w, w + W, ... and keeps per-column partial sums of
// dgamma / dbeta / (scaled dx column sum) in registers; one LDS reduction over the
// block's 4 waves, then the block stores its partial sums for the finalize kernel.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ dx, bf16* __restrict__ dxb,
                                                     const float* __restrict__ rowscale, int rows_per_scale,
                                                     float* __restrict__ partial, int M, int E) {
    __shared__ float red[3][4][64 * 4 + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = E >> 2;
    const int nwaves = gridDim.x * 4;
    f32x4 ag[MAXC], ab[MAXC], ac[MAXC], g[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        ag[i] = ab[i] = ac[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = lane + 64 * i;
        g[i] = c < nch ? *reinterpret_cast<const f32x4*>(gamma + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += nwaves) {
        const float mu = mean[row], rs = rstd[row];
        const float sc = rowscale ? rowscale[row / rows_per_scale] : 1.f;
        f32x4 xh[MAXC], gy[MAXC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)row * E + 4 * c);
                const bf16x4 dv = *reinterpret_cast<const bf16x4*>(dy + (size_t)row * E + 4 * c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = (float)dv[r];
                    xh[i][r] = (xv[r] - mu) * rs;
                    gy[i][r] = d * g[i][r];
                    s1 += gy[i][r];
                    s2 += gy[i][r] * xh[i][r];
                    ag[i][r] += d * xh[i][r];
                    ab[i][r] += d;
                }
            }
        }
        s1 = wave_sum(s1) / (float)E;
        s2 = wave_sum(s2) / (float)E;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = rs * (gy[i][r] - s1 - xh[i][r] * s2);
                if (dres) o += *reinterpret_cast<const f32x4*>(dres + (size_t)row * E + 4 * c);
                if (dx) *reinterpret_cast<f32x4*>(dx + (size_t)row * E + 4 * c) = o;
                o *= sc;
                ac[i] += o;
                if (dxb) *reinterpret_cast<bf16x4*>(dxb + (size_t)row * E + 4 * c) = pack4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    // cross-wave reduction, one chunk index at a time
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        if (64 * i >= nch) break;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[0][wave][lane * 4 + r] = ag[i][r];
            red[1][wave][lane * 4 + r] = ab[i][r];
            red[2][wave][lane * 4 + r] = ac[i][r];
        }
        __syncthreads();
        // 256 threads cover the 256 columns of this chunk group
        const int col = threadIdx.x;            // column within the group: lane' * 4 + r
        const int c = (col >> 2) + 64 * i;      // chunk index
        if (c < nch) {
            const int e = 4 * c + (col & 3);
            const float g_ = red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col];
            const float b_ = red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col];
            const float c_ = red[2][0][col] + red[2][1][col] + red[2][2][col] + red[2][3][col];
            // per-block partial sums [block][3][E]; hundreds of blocks adding atomically into the
            // same E addresses would serialise at the memory side
            float* pp = partial + (size_t)blockIdx.x * 3 * E;
            pp[e] = g_;
            pp[E + e] = b_;
            pp[2 * E + e] = c_;
        }
    }
}

// sum the per-block partials over blocks and accumulate into dgamma / dbeta / dcolsum
__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int E,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ dcolsum) {
    __shared__ float red[8][32];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;  // 32 columns x 8 block lanes
    const int j = blockIdx.x * 32 + cx;                      // index into the 3E partial columns
    float s = 0.f;
    if (j < 3 * E)
        for (int b = ry; b < nblocks; b += 8) s += partial[(size_t)b * 3 * E + j];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && j < 3 * E) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][cx];
        float* dst = j < E ? dgamma : (j < 2 * E ? dbeta : dcolsum);
        if (dst) dst[j % E] += t;
    }
}

}  // namespace

extern "C" int vp_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                                float* mean, float* rstd, int M, int E, float eps, hipStream_t stream) {
    VP_CHECK_ARG(x && gamma && beta && (y_bf16 || y_f32));
    VP_CHECK_ARG(M > 0 && E > 0 && E % 4 == 0 && E <= 256 * MAXC);
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, x, gamma, beta, (bf16*)y_bf16, y_f32,
                       mean, rstd, M, E, eps);
    VP_LAUNCH_RET();
}

extern "C" int vp_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                                const float* gamma, const float* dres, float* dx, void* dxb_bf16,
                                const float* rowscale, int rows_per_scale, float* dgamma, float* dbeta,
                                float* dcolsum, int M, int E, float* workspace, size_t workspace_floats,
                                hipStream_t stream) {
    VP_CHECK_ARG(dy_bf16 && x && mean && rstd && gamma && (dx || dxb_bf16) && workspace);
    VP_CHECK_ARG(M > 0 && E > 0 && E % 4 == 0 && E <= 256 * MAXC);
    if (rowscale) VP_CHECK_ARG(rows_per_scale > 0);
    int blocks = (M + 3) / 4;
    if (blocks > 512) blocks = 512;
    if ((size_t)blocks * 3 * E > workspace_floats) blocks = (int)(workspace_floats / (3 * (size_t)E));
    VP_CHECK_ARG(blocks >= 1);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16*)dy_bf16, x, mean, rstd, gamma,
                       dres, dx, (bf16*)dxb_bf16, rowscale, rows_per_scale, workspace, M, E);
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((3 * E + 31) / 32), dim3(256), 0, stream, workspace, blocks, E,
                       dgamma, dbeta, dcolsum);
    VP_LAUNCH_RET();
}
