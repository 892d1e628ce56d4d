// JointsMSELoss on NCHW heatmaps (the layout the reference's head returns): loss value and
// gradient in one streaming pass.  HBM-bound: reads pred + target, writes dpred.
#include "common.h"
#include "vitpose_hip.h"

namespace {

// one workgroup-row per (image, joint) plane segment: the joint weight is uniform per block row
__global__ __launch_bounds__(256) void mse_loss_nchw_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ target,
                                                            const float* __restrict__ tw, float* __restrict__ loss,
                                                            float* __restrict__ dpred, int HW, float coef) {
    const int plane = blockIdx.y;  // b * K + k
    const float w = tw ? tw[plane] : 1.f;
    const size_t base = (size_t)plane * HW;
    float part = 0.f;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < HW; i += gridDim.x * 256 * 4) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(pred + base + i);
        const f32x4 t = *reinterpret_cast<const f32x4*>(target + base + i);
        f32x4 d = (p - t) * w;
        part += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        if (dpred) *reinterpret_cast<f32x4*>(dpred + base + i) = d * (2.f * w * coef);
    }
    part = wave_sum(part);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * coef);
}

}  // namespace

extern "C" int vp_mse_loss_nchw(const float* pred, const float* target, const float* tw, float* loss, float* dpred,
                                int B, int K, int HW, float loss_weight, hipStream_t stream) {
    VP_CHECK_ARG(pred && target && loss && B > 0 && K > 0 && HW > 0 && HW % 4 == 0);
    const float coef = loss_weight / ((float)B * (float)K * (float)HW);
    int gx = (HW / 4 + 255) / 256;
    if (gx > 8) gx = 8;
    hipLaunchKernelGGL(mse_loss_nchw_kernel, dim3(gx, B * K), dim3(256), 0, stream, pred, target, tw, loss, dpred, HW,
                       coef);
    VP_LAUNCH_RET();
}
