// Input side of the top-down pipeline on the device: crop-and-warp of the person box + tensor
// normalisation, and the heat-map targets.  The reference does these per sample on CPU workers
// (cv2.warpAffine, torchvision to_tensor / normalize, numpy target painting).
#include "common.h"
#include "vitpose_hip.h"

namespace {

struct Norm3 {
    float mean[3], std[3];
};

// cv2.warpAffine(src, M, (Wo, Ho), flags=INTER_LINEAR) on 8-bit RGB, restated with OpenCV's fixed-point
// scheme: source coordinates in 1/1024 px from the inverted matrix (AB_BITS = 10), reduced to 1/32 px
// (INTER_BITS = 5), bilinear weights (32-a)(32-b)*32 of 32768, result (sum + 16384) >> 15, taps outside
// the image read the constant border 0.  Followed by to_tensor (/255) and normalize ((v - mean) / std).
__global__ __launch_bounds__(256) void warp_affine_norm_kernel(const unsigned char* __restrict__ imgs,
                                                               const long* __restrict__ offs,
                                                               const int* __restrict__ hw,
                                                               const double* __restrict__ minv,
                                                               float* __restrict__ out, int Ho, int Wo, Norm3 nm) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Ho * Wo) return;
    const int x = p % Wo, y = p / Wo;
    const int H = hw[2 * b], W = hw[2 * b + 1];
    const double* m = minv + 6 * b;
    const unsigned char* src = imgs + offs[b];
    const int AB = 1024, RD = 16;
    const int adx = __double2int_rn(m[0] * x * AB), bdx = __double2int_rn(m[3] * x * AB);
    const int X0 = __double2int_rn((m[1] * y + m[2]) * AB) + RD, Y0 = __double2int_rn((m[4] * y + m[5]) * AB) + RD;
    const int X = (X0 + adx) >> 5, Y = (Y0 + bdx) >> 5;
    int sx = X >> 5, sy = Y >> 5;
    sx = max(-32768, min(32767, sx));
    sy = max(-32768, min(32767, sy));
    const int ax = X & 31, ay = Y & 31;
    const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
    const bool x0ok = sx >= 0 && sx < W, x1ok = sx + 1 >= 0 && sx + 1 < W;
    const bool y0ok = sy >= 0 && sy < H, y1ok = sy + 1 >= 0 && sy + 1 < H;
    const size_t HWo = (size_t)Ho * Wo;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int acc = 0;
        if (y0ok && x0ok) acc += w00 * src[((size_t)sy * W + sx) * 3 + c];
        if (y0ok && x1ok) acc += w01 * src[((size_t)sy * W + sx + 1) * 3 + c];
        if (y1ok && x0ok) acc += w10 * src[((size_t)(sy + 1) * W + sx) * 3 + c];
        if (y1ok && x1ok) acc += w11 * src[((size_t)(sy + 1) * W + sx + 1) * 3 + c];
        const int v = (acc + 16384) >> 15;
        out[((size_t)b * 3 + c) * HWo + p] = ((float)v / 255.0f - nm.mean[c]) / nm.std[c];
    }
}

// One workgroup per (sample, joint).  encoding 0: UDP Gaussian (_udp_generate_target), 1: MSRA
// (_msra_generate_target), 2: MSRA with unbiased encoding.
__global__ __launch_bounds__(256) void generate_target_kernel(const float* __restrict__ joints,
                                                              const float* __restrict__ visible,
                                                              const float* __restrict__ joint_weights,
                                                              float* __restrict__ target, float* __restrict__ tw,
                                                              int K, int H, int W, int img_w, int img_h, double sigma,
                                                              int encoding) {
    const int bk = blockIdx.x, k = bk % K;
    const double jx = joints[3 * (size_t)bk], jy = joints[3 * (size_t)bk + 1];
    float* tgt = target + (size_t)bk * H * W;
    double weight = visible[bk];
    const double tmp = sigma * 3;
    const int size = (int)(2 * tmp + 1);
    double fsx, fsy;
    if (encoding == 0) {
        fsx = (img_w - 1.0) / (W - 1.0);
        fsy = (img_h - 1.0) / (H - 1.0);
    } else {
        fsx = (double)img_w / W;
        fsy = (double)img_h / H;
    }
    const double mxa = jx / fsx, mya = jy / fsy;
    int ulx, uly, brx, bry, mux = 0, muy = 0;
    bool inside;
    if (encoding == 2) {
        inside = !(mxa - tmp >= W || mya - tmp >= H || mxa + tmp + 1 < 0 || mya + tmp + 1 < 0);
        ulx = uly = brx = bry = 0;
    } else {
        mux = (int)(mxa + 0.5);
        muy = (int)(mya + 0.5);
        ulx = (int)(mux - tmp);
        uly = (int)(muy - tmp);
        brx = (int)(mux + tmp + 1);
        bry = (int)(muy + tmp + 1);
        inside = !(ulx >= W || uly >= H || brx < 0 || bry < 0);
    }
    if (!inside) weight = 0;
    const bool paint = weight > 0.5;
    double x0 = size / 2, y0 = size / 2;
    if (encoding == 0) {
        x0 += mxa - mux;
        y0 += mya - muy;
    }
    const double inv2s2 = 1.0 / (2 * sigma * sigma);
    for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
        const int x = i % W, y = i / W;
        float v = 0.f;
        if (paint) {
            if (encoding == 2) {
                const float dx = (float)x - (float)mxa, dy = (float)y - (float)mya;   // f32 grids in the reference
                v = (float)exp(-((double)(dx * dx) + (double)(dy * dy)) * inv2s2);
            } else if (x >= ulx && x < brx && y >= uly && y < bry) {
                const double gx = (x - ulx) - x0, gy = (y - uly) - y0;
                v = (float)exp(-(gx * gx + gy * gy) * inv2s2);
            }
        }
        tgt[i] = v;
    }
    if (threadIdx.x == 0) tw[bk] = (float)(joint_weights ? weight * joint_weights[k] : weight);
}

}  // namespace

extern "C" int vp_warp_affine_normalize(const unsigned char* imgs, const long* offsets, const int* hw,
                                        const double* inv_matrices, float* out, int B, int Ho, int Wo,
                                        const float mean[3], const float std[3], hipStream_t stream) {
    VP_CHECK_ARG(imgs && offsets && hw && inv_matrices && out && B > 0 && Ho > 0 && Wo > 0 && mean && std);
    Norm3 nm;
    for (int c = 0; c < 3; ++c) {
        nm.mean[c] = mean[c];
        nm.std[c] = std[c];
        VP_CHECK_ARG(std[c] != 0.f);
    }
    dim3 grid((Ho * Wo + 255) / 256, B);
    hipLaunchKernelGGL(warp_affine_norm_kernel, grid, dim3(256), 0, stream, imgs, offsets, hw, inv_matrices, out, Ho,
                       Wo, nm);
    VP_LAUNCH_RET();
}

extern "C" int vp_generate_target(const float* joints, const float* visible, const float* joint_weights,
                                  float* target, float* target_weight, int B, int K, int H, int W, int img_w,
                                  int img_h, float sigma, int encoding, hipStream_t stream) {
    VP_CHECK_ARG(joints && visible && target && target_weight && B > 0 && K > 0 && H > 1 && W > 1);
    VP_CHECK_ARG(img_w > 1 && img_h > 1 && sigma > 0.f && encoding >= 0 && encoding <= 2);
    hipLaunchKernelGGL(generate_target_kernel, dim3(B * K), dim3(256), 0, stream, joints, visible, joint_weights,
                       target, target_weight, K, H, W, img_w, img_h, (double)sigma, encoding);
    VP_LAUNCH_RET();
}
