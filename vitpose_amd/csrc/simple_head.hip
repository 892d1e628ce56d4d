// "Simple decoder" of the top-down heatmap head: ReLU -> bilinear upsample -> Conv2d(k x k),
// plus the NCHW <-> channels-last bridges of the stand-alone head / backbone modules.
//
// The conv and the upsample are both linear, so the channel contraction is done FIRST, on the
// low-resolution token grid, by the MFMA GEMM:  G[(b,j)][t*KP + k] = sum_c W[k][c][t] * relu(f)[b][j][c]
// (t = conv tap).  What remains is a per-pixel interpolate-shift-sum over the k*k tap planes of G,
//   out[b][k][y][x] = bias[k] + sum_t Up(G_t)[y + ty - p][x + tx - p][k]      (0 outside the up-sampled grid),
// which never materialises the up-sampled [B][C][sH][sW] feature map (7.5 MB per image for ViT-B).
// Its transpose (backward) gathers dOut into dG; the weight / data gradients are GEMMs again.
// Interpolation tables (source index pairs and weights per output coordinate, and their transposes)
// are built on the host, so any scale factor and align_corners setting is covered.
#include "common.h"
#include "vitpose_hip.h"

namespace {

inline int grid_for(size_t work_items, int cap = 8192) {
    size_t g = (work_items + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g ? g : 1));
}

// One thread per output pixel (b, y, x): K accumulators, x fastest so the NCHW stores coalesce.
template <int KP4>
__global__ __launch_bounds__(256) void upconv_fwd_kernel(const float* __restrict__ G, int ldg,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         const int* __restrict__ yi, const float* __restrict__ yw,
                                                         const int* __restrict__ xi, const float* __restrict__ xw,
                                                         int B, int K, int Hg, int Wg, int Ho, int Wo, int ks) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * Ho * Wo) return;
    const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), b = (int)(i / ((size_t)Wo * Ho));
    constexpr int KP = KP4 * 4;
    f32x4 acc[KP4];
#pragma unroll
    for (int q = 0; q < KP4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pad = (ks - 1) / 2;
    const float* Gb = G + (size_t)b * Hg * Wg * ldg;
    for (int ty = 0; ty < ks; ++ty) {
        const int uy = y + ty - pad;
        if (uy < 0 || uy >= Ho) continue;
        for (int tx = 0; tx < ks; ++tx) {
            const int ux = x + tx - pad;
            if (ux < 0 || ux >= Wo) continue;
            const int t = ty * ks + tx;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float w = yw[2 * uy + a] * xw[2 * ux + c];
                    const float* g = Gb + (size_t)(yi[2 * uy + a] * Wg + xi[2 * ux + c]) * ldg + t * KP;
#pragma unroll
                    for (int q = 0; q < KP4; ++q) acc[q] += *reinterpret_cast<const f32x4*>(g + 4 * q) * w;
                }
        }
    }
    float* o = out + ((size_t)b * K * Ho + y) * Wo + x;
#pragma unroll
    for (int q = 0; q < KP4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * q + r;
            if (k < K) o[(size_t)k * Ho * Wo] = acc[q][r] + (bias ? bias[k] : 0.f);
        }
}

// One thread per (b, jy, tap, jx): dG[(b,j)][t*KP + k] = sum over the up-sampled positions u that
// interpolate from j, of  c(u,j) * dOut[b][k][u - (t - p)].  tyi/tyw [Hg][L]: the u's of row jy and
// their weights (index -1 = unused slot), txi/txw likewise for columns.
template <int KP4>
__global__ __launch_bounds__(256) void upconv_bwd_kernel(const float* __restrict__ dout, bf16* __restrict__ dG,
                                                         int ldg, const int* __restrict__ tyi,
                                                         const float* __restrict__ tyw, const int* __restrict__ txi,
                                                         const float* __restrict__ txw, int Ly, int Lx, int B, int K,
                                                         int Hg, int Wg, int Ho, int Wo, int ks) {
    const int taps = ks * ks;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * Hg * taps * Wg) return;
    const int jx = (int)(i % Wg), t = (int)((i / Wg) % taps), jy = (int)((i / ((size_t)Wg * taps)) % Hg);
    const int b = (int)(i / ((size_t)Wg * taps * Hg));
    constexpr int KP = KP4 * 4;
    const int pad = (ks - 1) / 2, ty = t / ks, tx = t % ks;
    float acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k] = 0.f;
    const float* db = dout + (size_t)b * K * Ho * Wo;
    for (int a = 0; a < Ly; ++a) {
        const int uy = tyi[jy * Ly + a];
        const int y = uy - ty + pad;
        if (uy < 0 || y < 0 || y >= Ho) continue;
        const float wy = tyw[jy * Ly + a];
        for (int c = 0; c < Lx; ++c) {
            const int ux = txi[jx * Lx + c];
            const int x = ux - tx + pad;
            if (ux < 0 || x < 0 || x >= Wo) continue;
            const float w = wy * txw[jx * Lx + c];
            const float* d = db + (size_t)y * Wo + x;
#pragma unroll
            for (int k = 0; k < KP; ++k)
                if (k < K) acc[k] += w * d[(size_t)k * Ho * Wo];
        }
    }
    bf16* o = dG + ((size_t)(b * Hg + jy) * Wg + jx) * ldg;
#pragma unroll
    for (int q = 0; q < KP4; ++q)
        *reinterpret_cast<bf16x4*>(o + t * KP + 4 * q) = pack4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    if (t == taps - 1)
        for (int k = taps * KP; k < ldg; k += 4) *reinterpret_cast<bf16x4*>(o + k) = pack4(0.f, 0.f, 0.f, 0.f);
}

// out[k] += sum_b sum_p x[b][k][p]: one workgroup per (k, b) plane
__global__ __launch_bounds__(256) void plane_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int K,
                                                        int HW) {
    const float* p = x + ((size_t)blockIdx.y * K + blockIdx.x) * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) s += p[i];
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + blockIdx.x, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void relu_bf16_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (float)v[j] > 0.f ? v[j] : (bf16)0.f;
        *reinterpret_cast<bf16x8*>(y + i * 8) = o;
    }
}

// NCHW f32 [B][C][HW] -> channels-last bf16 [B*HW][ld], columns >= C zeroed.  One thread per
// (b, 8-channel chunk, pixel), pixel fastest: the 8 plane reads coalesce, each lane stores 16 B.
__global__ __launch_bounds__(256) void nchw_to_cl_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int B,
                                                         int C, int HW, int ld) {
    const int chunks = ld >> 3;
    const size_t total = (size_t)B * chunks * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW), ch = (int)((i / HW) % chunks), b = (int)(i / ((size_t)HW * chunks));
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ch * 8 + j;
            o[j] = (bf16)(c < C ? src[((size_t)b * C + c) * HW + p] : 0.f);
        }
        *reinterpret_cast<bf16x8*>(dst + ((size_t)b * HW + p) * ld + ch * 8) = o;
    }
}

// 'resize_concat' input bridge: one NCHW f32 map [B][C][Hi][Wi], bilinearly resized to Ho x Wo, into the columns
// [coff, coff + C) of the channels-last bf16 matrix [B*Ho*Wo][ld].  yi/yw [Ho][2], xi/xw [Wo][2]: source index pairs
// and weights per output coordinate.  One thread per (b, 8-channel chunk, output pixel), pixel fastest.
__global__ __launch_bounds__(256) void resize_to_cl_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                           const int* __restrict__ yi, const float* __restrict__ yw,
                                                           const int* __restrict__ xi, const float* __restrict__ xw,
                                                           int B, int C, int Hi, int Wi, int Ho, int Wo, int ld, int coff) {
    const int chunks = C >> 3, HWo = Ho * Wo;
    const size_t total = (size_t)B * chunks * HWo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HWo), ch = (int)((i / HWo) % chunks), b = (int)(i / ((size_t)HWo * chunks));
        const int y = p / Wo, x = p % Wo;
        const int y0 = yi[2 * y], y1 = yi[2 * y + 1], x0 = xi[2 * x], x1 = xi[2 * x + 1];
        const float w00 = yw[2 * y] * xw[2 * x], w01 = yw[2 * y] * xw[2 * x + 1];
        const float w10 = yw[2 * y + 1] * xw[2 * x], w11 = yw[2 * y + 1] * xw[2 * x + 1];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* s = src + ((size_t)b * C + ch * 8 + j) * Hi * Wi;
            o[j] = (bf16)(w00 * s[y0 * Wi + x0] + w01 * s[y0 * Wi + x1] + w10 * s[y1 * Wi + x0] + w11 * s[y1 * Wi + x1]);
        }
        *reinterpret_cast<bf16x8*>(dst + ((size_t)b * HWo + p) * ld + coff + ch * 8) = o;
    }
}

// Its transpose: dx[b][c][j][i] = sum over the output pixels (u, v) that interpolate from (j, i) of
// wy * wx * g[(b, u, v)][coff + c].  tyi/tyw [Hi][Ly], txi/txw [Wi][Lx] list those pixels (-1 = unused slot).
// One thread per (b, 8-channel chunk, source pixel), pixel fastest: the eight plane stores coalesce.
__global__ __launch_bounds__(256) void resize_to_cl_bwd_kernel(const bf16* __restrict__ g, float* __restrict__ dx,
                                                               const int* __restrict__ tyi, const float* __restrict__ tyw,
                                                               const int* __restrict__ txi, const float* __restrict__ txw,
                                                               int Ly, int Lx, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                                               int ld, int coff) {
    const int chunks = C >> 3, HWi = Hi * Wi;
    const size_t total = (size_t)B * chunks * HWi;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(t % HWi), ch = (int)((t / HWi) % chunks), b = (int)(t / ((size_t)HWi * chunks));
        const int j = p / Wi, i = p % Wi;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < Ly; ++a) {
            const int u = tyi[j * Ly + a];
            if (u < 0) continue;
            for (int c = 0; c < Lx; ++c) {
                const int v = txi[i * Lx + c];
                if (v < 0) continue;
                const float w = tyw[j * Ly + a] * txw[i * Lx + c];
                const bf16x8 q = *reinterpret_cast<const bf16x8*>(g + ((size_t)(b * Ho + u) * Wo + v) * ld + coff + ch * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += w * (float)q[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) dx[((size_t)b * C + ch * 8 + k) * HWi + p] = acc[k];
    }
}

}  // namespace

extern "C" int vp_resize_to_cl_bf16(const float* src, void* dst, const int* yi, const float* yw, const int* xi,
                                    const float* xw, int B, int C, int Hi, int Wi, int Ho, int Wo, int ld, int coff,
                                    hipStream_t stream) {
    VP_CHECK_ARG(src && dst && yi && yw && xi && xw && B > 0 && C > 0 && C % 8 == 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
    VP_CHECK_ARG(coff >= 0 && coff % 8 == 0 && ld % 8 == 0 && coff + C <= ld);
    const size_t total = (size_t)B * (C / 8) * Ho * Wo;
    hipLaunchKernelGGL(resize_to_cl_kernel, dim3(grid_for(total)), dim3(256), 0, stream, src, (bf16*)dst, yi, yw, xi, xw, B,
                       C, Hi, Wi, Ho, Wo, ld, coff);
    VP_LAUNCH_RET();
}

extern "C" int vp_resize_to_cl_bwd(const void* g, float* dx, const int* tyi, const float* tyw, const int* txi,
                                   const float* txw, int Ly, int Lx, int B, int C, int Hi, int Wi, int Ho, int Wo, int ld,
                                   int coff, hipStream_t stream) {
    VP_CHECK_ARG(g && dx && tyi && tyw && txi && txw && Ly > 0 && Lx > 0 && B > 0 && C > 0 && C % 8 == 0);
    VP_CHECK_ARG(Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && coff >= 0 && coff % 8 == 0 && ld % 8 == 0 && coff + C <= ld);
    const size_t total = (size_t)B * (C / 8) * Hi * Wi;
    hipLaunchKernelGGL(resize_to_cl_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16*)g, dx, tyi, tyw,
                       txi, txw, Ly, Lx, B, C, Hi, Wi, Ho, Wo, ld, coff);
    VP_LAUNCH_RET();
}

extern "C" int vp_upconv_fwd(const float* G, int ldg, const float* bias, float* out, const int* yi, const float* yw,
                             const int* xi, const float* xw, int B, int K, int kpad, int Hg, int Wg, int Ho, int Wo,
                             int ksize, hipStream_t stream) {
    VP_CHECK_ARG(G && out && yi && yw && xi && xw && B > 0 && K > 0 && K <= kpad && Hg > 0 && Wg > 0 && Ho > 0 && Wo > 0);
    VP_CHECK_ARG((ksize == 1 || ksize == 3) && ldg >= ksize * ksize * kpad && ldg % 4 == 0);
    const size_t n = (size_t)B * Ho * Wo;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define VP_UF(KP4)                                                                                               \
    hipLaunchKernelGGL(upconv_fwd_kernel<KP4>, grid, block, 0, stream, G, ldg, bias, out, yi, yw, xi, xw, B, K, \
                       Hg, Wg, Ho, Wo, ksize)
    switch (kpad) {
        case 20: VP_UF(5); break;
        case 32: VP_UF(8); break;
        case 64: VP_UF(16); break;
        default: return VP_ERR_UNSUPPORTED;
    }
#undef VP_UF
    VP_LAUNCH_RET();
}

extern "C" int vp_upconv_bwd(const float* dout, void* dG, int ldg, const int* tyi, const float* tyw, const int* txi,
                             const float* txw, int Ly, int Lx, int B, int K, int kpad, int Hg, int Wg, int Ho, int Wo,
                             int ksize, hipStream_t stream) {
    VP_CHECK_ARG(dout && dG && tyi && tyw && txi && txw && Ly > 0 && Lx > 0 && B > 0 && K > 0 && K <= kpad);
    VP_CHECK_ARG((ksize == 1 || ksize == 3) && ldg >= ksize * ksize * kpad && ldg % 4 == 0);
    const size_t n = (size_t)B * Hg * Wg * ksize * ksize;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define VP_UB(KP4)                                                                                                \
    hipLaunchKernelGGL(upconv_bwd_kernel<KP4>, grid, block, 0, stream, dout, (bf16*)dG, ldg, tyi, tyw, txi, txw, \
                       Ly, Lx, B, K, Hg, Wg, Ho, Wo, ksize)
    switch (kpad) {
        case 20: VP_UB(5); break;
        case 32: VP_UB(8); break;
        case 64: VP_UB(16); break;
        default: return VP_ERR_UNSUPPORTED;
    }
#undef VP_UB
    VP_LAUNCH_RET();
}

extern "C" int vp_plane_sum_f32(const float* x, float* out, int B, int K, int HW, hipStream_t stream) {
    VP_CHECK_ARG(x && out && B > 0 && K > 0 && HW > 0);
    hipLaunchKernelGGL(plane_sum_kernel, dim3(K, B), dim3(256), 0, stream, x, out, K, HW);
    VP_LAUNCH_RET();
}

extern "C" int vp_relu_bf16(const void* x, void* y, size_t n, hipStream_t stream) {
    VP_CHECK_ARG(x && y && n > 0 && n % 8 == 0);
    hipLaunchKernelGGL(relu_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, n / 8);
    VP_LAUNCH_RET();
}

extern "C" int vp_nchw_to_cl_bf16(const float* src, void* dst, int B, int C, int HW, int ld, hipStream_t stream) {
    VP_CHECK_ARG(src && dst && B > 0 && C > 0 && HW > 0 && ld >= C && ld % 8 == 0);
    const size_t total = (size_t)B * (ld / 8) * HW;
    hipLaunchKernelGGL(nchw_to_cl_kernel, dim3(grid_for(total)), dim3(256), 0, stream, src, (bf16*)dst, B, C, HW, ld);
    VP_LAUNCH_RET();
}
