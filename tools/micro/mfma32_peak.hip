// Sustained fp32 MFMA throughput, 16x16x4 against 32x32x2 (registers only), and the 32x32x2 loop with one ds_read_b32 pair per MFMA.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma32_peak mfma32_peak.hip ; run: ./mfma32_peak [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;       // waves per SIMD
    const int blocks = 256 * wps, iters = 20000;
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("waves/SIMD %d: 16x16x4 x16 acc %.3f ms  %.1f TFLOP/s\n", wps, ms, (double)blocks * 4 * iters * 16 * 2048.0 / ms / 1e9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k32<6>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("waves/SIMD %d: 32x32x2 x6 acc  %.3f ms  %.1f TFLOP/s\n", wps, ms, (double)blocks * 4 * iters * 6 * 4096.0 / ms / 1e9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k32<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("waves/SIMD %d: 32x32x2 x3 acc  %.3f ms  %.1f TFLOP/s\n", wps, ms, (double)blocks * 4 * iters * 3 * 4096.0 / ms / 1e9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("waves/SIMD %d: 32x32x2 x2 acc  %.3f ms  %.1f TFLOP/s\n", wps, ms, (double)blocks * 4 * iters * 2 * 4096.0 / ms / 1e9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("waves/SIMD %d: 32x32x2 x1 acc  %.3f ms  %.1f TFLOP/s\n", wps, ms, (double)blocks * 4 * iters * 1 * 4096.0 / ms / 1e9);
    }
    return 0;
}
