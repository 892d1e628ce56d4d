// What costs an LDS-fed MFMA loop its rate?  Same MFMA stream as mfma_peak_probe (v_mfma_f32_16x16x32_bf16, 8 waves per
// CU, random data), with the fragment traffic of the 192 x 192 GEMM tile added step by step:
//   mode 0: register operands only
//   mode 1: + 9 ds_read_b128 per 18 MFMAs (the fragment reads of one phase), no barrier
//   mode 2: + two s_barrier per 18 MFMAs (the phase structure of the ping-pong kernels)
//   mode 3: mode 2 with waves 4-7 one barrier interval behind waves 0-3 (reads of one group under the MFMAs of the other)
//   mode 4: mode 3 + 3 LDS-DMA loads (1 KiB each) per wave and phase from an L2-resident buffer, counted vmcnt wait:
//           the operand traffic of the 192 x 192 tile (48 KiB per two phases and workgroup)
//   mode 5: mode 4 with the source in a 2 GiB buffer walked once (HBM instead of L2)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds_probe mfma_lds_probe.hip && ./mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const bf16x8* __restrict__ in, float* out, int iters, const char* gsrc,
                                                size_t gspan) {
    __shared__ __attribute__((aligned(16))) char lds[147456];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 147456 / 16; i += 512) reinterpret_cast<bf16x8*>(lds)[i] = in[i & 4095];
    __syncthreads();
    bf16x8 a[6], b[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = in[(tid * 9 + i) & 4095];
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = in[(tid * 9 + 6 + i) & 4095];
    f32x4 acc[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* base = lds + (wave & 3) * 9216 + lane * 16;   // 9 fragments of 1 KiB per wave and phase, conflict-free
    if (MODE >= 3 && wave >= 4) __builtin_amdgcn_s_barrier();
    size_t goff = ((size_t)blockIdx.x * 8 + wave) * 3072 + lane * 16;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const char* p = base + (it & 3) * 36864;
        if constexpr (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 6; ++i) a[i] = *reinterpret_cast<const bf16x8*>(p + i * 1024);
#pragma unroll
            for (int i = 0; i < 3; ++i) b[i] = *reinterpret_cast<const bf16x8*>(p + (6 + i) * 1024);
        }
        if constexpr (MODE >= 4) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + goff + k * 1024),
                                                 (__attribute__((address_space(3))) void*)(lds + ((it + 2) % 3) * 49152 + (wave * 3 + k) * 1024), 16, 0, 0);
            }
            goff += (size_t)256 * 8 * 3072;
            if (goff >= gspan) goff -= gspan;
            if (it & 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        if constexpr (MODE >= 2) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i + 3 * s][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i + 3 * s], acc[i + 3 * s][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (MODE >= 2) __builtin_amdgcn_s_barrier();
    }
    if (MODE >= 3 && wave < 4) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) sum += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = sum;
}

// mode 6: the same 192 x 192 x 64 tile with the K tile split between the wave groups: wave (r, c, h) owns the 96 x 96 block
// (r, c) for the 32-wide sub-step h -- 12 fragment reads and 6 LDS-DMA loads per 36 MFMAs and TWO barriers per 36 MFMAs
// (a third fewer LDS read bytes, half the barriers; the two halves would be summed once per output tile).
__global__ __launch_bounds__(512, 2) void probe_ksplit(const bf16x8* __restrict__ in, float* out, int iters, const char* gsrc,
                                                       size_t gspan) {
    __shared__ __attribute__((aligned(16))) char lds[147456];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 147456 / 16; i += 512) reinterpret_cast<bf16x8*>(lds)[i] = in[i & 4095];
    __syncthreads();
    bf16x8 a[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { a[i] = in[(tid * 12 + i) & 4095]; b[i] = in[(tid * 12 + 6 + i) & 4095]; }
    f32x4 acc[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* base = lds + (wave & 3) * 12288 + lane * 16;
    if (wave >= 4) __builtin_amdgcn_s_barrier();
    size_t goff = ((size_t)blockIdx.x * 8 + wave) * 6144 + lane * 16;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const char* p = base + (it % 3) * 49152;
#pragma unroll
        for (int i = 0; i < 6; ++i) a[i] = *reinterpret_cast<const bf16x8*>(p + i * 1024);
#pragma unroll
        for (int i = 0; i < 6; ++i) b[i] = *reinterpret_cast<const bf16x8*>(p + (6 + i) * 1024);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + goff + k * 1024),
                                             (__attribute__((address_space(3))) void*)(lds + ((it + 2) % 3) * 49152 + (wave * 6 + k) * 1024), 16, 0, 0);
        }
        goff += (size_t)256 * 8 * 6144;
        if (goff >= gspan) goff -= gspan;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // the loads of the K tile that is read next
        __builtin_amdgcn_s_barrier();
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) sum += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = sum;
}

// mode 7: the 96 x 48 wave tile of mode 4 with BOTH sub-steps of a K tile in one interval: 18 fragment reads and 6 LDS-DMA
// loads per 36 MFMAs, two barriers per 36 MFMAs (same LDS traffic as mode 4, half the barriers, intervals twice as long).
__global__ __launch_bounds__(512, 2) void probe_long(const bf16x8* __restrict__ in, float* out, int iters, const char* gsrc,
                                                     size_t gspan) {
    __shared__ __attribute__((aligned(16))) char lds[147456];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 147456 / 16; i += 512) reinterpret_cast<bf16x8*>(lds)[i] = in[i & 4095];
    __syncthreads();
    bf16x8 a[6][2], b[3][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a[i][s] = in[(tid * 18 + s * 9 + i) & 4095];
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i][s] = in[(tid * 18 + s * 9 + 6 + i) & 4095];
    }
    f32x4 acc[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* base = lds + (wave & 3) * 9216 + lane * 16;
    if (wave >= 4) __builtin_amdgcn_s_barrier();
    size_t goff = ((size_t)blockIdx.x * 8 + wave) * 6144 + lane * 16;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const char* p = base + (it % 3) * 49152;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < 6; ++i) a[i][s] = *reinterpret_cast<const bf16x8*>(p + s * 36864 % 40000 + i * 1024);
#pragma unroll
            for (int i = 0; i < 3; ++i) b[i][s] = *reinterpret_cast<const bf16x8*>(p + s * 36864 % 40000 + (6 + i) * 1024);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + goff + k * 1024),
                                             (__attribute__((address_space(3))) void*)(lds + ((it + 2) % 3) * 49152 + (wave * 6 + k) * 1024), 16, 0, 0);
        }
        goff += (size_t)256 * 8 * 6144;
        if (goff >= gspan) goff -= gspan;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][s], a[i][s], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) sum += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = sum;
}

int main() {
    const int n = 4096 * 8;
    std::vector<unsigned short> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        float v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
        unsigned u; std::memcpy(&u, &v, 4);
        h[i] = (unsigned short)(u >> 16);
    }
    void *din, *dout;
    void* gbig;
    const size_t big = (size_t)2 << 30, l2span = (size_t)256 * 8 * 3072 * 4;   // 24 MiB: every workgroup re-reads 4 tiles
    if (hipMalloc(&din, n * 2) != hipSuccess || hipMalloc(&dout, 1024 * 512 * 4) != hipSuccess || hipMalloc(&gbig, big) != hipSuccess) return 1;
    hipMemset(gbig, 0x3c, big);
    hipMemcpy(din, h.data(), n * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256;
    for (int mode = 0; mode < 8; ++mode) {
        float best = 1e30f, last = 0.f;
        for (int rep = 0; rep < 60; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: probe<0><<<grid, 512>>>((const bf16x8*)din, (float*)dout, iters, (const char*)gbig, l2span); break;
                case 1: probe<1><<<grid, 512>>>((const bf16x8*)din, (float*)dout, iters, (const char*)gbig, l2span); break;
                case 2: probe<2><<<grid, 512>>>((const bf16x8*)din, (float*)dout, iters, (const char*)gbig, l2span); break;
                case 3: probe<3><<<grid, 512>>>((const bf16x8*)din, (float*)dout, iters, (const char*)gbig, l2span); break;
                case 4: probe<4><<<grid, 512>>>((const bf16x8*)din, (float*)dout, iters, (const char*)gbig, l2span); break;
                case 5: probe<4><<<grid, 512>>>((const bf16x8*)din, (float*)dout, iters, (const char*)gbig, big - ((size_t)64 << 20)); break;
                case 6: probe_ksplit<<<grid, 512>>>((const bf16x8*)din, (float*)dout, iters / 2, (const char*)gbig, l2span); break;
                default: probe_long<<<grid, 512>>>((const bf16x8*)din, (float*)dout, iters / 2, (const char*)gbig, l2span); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            last = ms; if (ms < best) best = ms;
        }
        const double flop = (double)grid * 8 * iters * 18 * 16384.0;
        printf("mode %d: last %.2f ms %.0f TFLOP/s (best %.0f)\n", mode, last, flop / last / 1e9, flop / best / 1e9);
    }
    return 0;
}
