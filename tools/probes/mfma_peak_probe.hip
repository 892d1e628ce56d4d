// Bare MFMA issue-rate probe for gfx950: register operands only, random bf16 data, 2 waves per SIMD.
// Compares v_mfma_f32_16x16x32_bf16 with v_mfma_f32_32x32x16_bf16 (same nominal rate, half the operand
// register reads per flop) to see what the silicon sustains under load.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak_probe mfma_peak_probe.hip && ./mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void probe(const bf16x8* __restrict__ in, float* out, int iters, long long* clk) {
    const int tid = threadIdx.x;
    const long long c0 = clock64(), w0 = wall_clock64();   // shader clock / constant 100 MHz counter
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(tid * 8 + i) & 4095];
        b[i] = in[(tid * 8 + 4 + i) & 4095];
    }
    float sum = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * rep], b[j + 2 * rep], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][15];
    }
    out[blockIdx.x * 512 + tid] = sum;
    if (blockIdx.x == 0 && tid == 0) {
        clk[0] = clock64() - c0;
        clk[1] = wall_clock64() - w0;
    }
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
    hipMalloc(&din, n * 2);
    hipMalloc(&dout, 1024 * 512 * 4);
    long long* dclk;
    hipMalloc((void**)&dclk, 16);
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero) {
        if (zero) hipMemset(din, 0, n * 2); else hipMemcpy(din, h.data(), n * 2, hipMemcpyHostToDevice);
        for (int shape : {16, 32}) {
            for (int rep = 0; rep < 240; ++rep) {
                const int iters = 40000, grid = 256;
                hipEventRecord(e0);
                if (shape == 16) probe<16><<<grid, 512>>>((const bf16x8*)din, (float*)dout, iters, dclk);
                else probe<32><<<grid, 512>>>((const bf16x8*)din, (float*)dout, iters, dclk);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                // per wave per iter: 16 MFMAs of 16384 flop (16x16x32) or 8 of 32768 (32x32x16)
                double flop = (double)grid * 8 * iters * 16 * 16384.0;
                if (rep % 40 == 39 || rep < 2) {
                    long long hc[2];
                    hipMemcpy(hc, dclk, 16, hipMemcpyDeviceToHost);
                    // core clock the loop ran at = shader-clock ticks per tick of the constant-rate counter
                    printf("%s data, %dx%d rep %d: %.2f ms  %.0f TFLOP/s  core clock %.0f MHz\n", zero ? "zero" : "random", shape,
                           shape, rep, ms, flop / ms / 1e9, (double)hc[0] / (double)hc[1] * wall_khz / 1e3);
                }
            }
        }
    }
    return 0;
}
