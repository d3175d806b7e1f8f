// Times the product NN GEMM kernel (eeg_gnn_ssl_amd/csrc/kernels_gemm.h) in isolation at the cfg2 shapes.
// (Round-1 experiments with persistent / 64- and 96-row-tile register-staged variants: DESIGN.md section 4.2.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../eeg_gnn_ssl_amd/csrc/kernels_gemm.h"
using namespace eeg;
static float* A; static float* Bp; static float* C; static float* bias;
template <typename K, typename... Args>
void timeit(const char* what, double flops, K kern, dim3 grid, size_t lds, Args... args) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, args...);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    printf("%-46s %.3f ms  %6.1f TFLOP/s %s\n", what, best, flops / best / 1e9, e == hipSuccess ? "" : hipGetErrorString(e));
}
template <int KC, int RTW> size_t lds3() { return 2 * (size_t)(32 * RTW * lds_stride(KC) + (KC / 4) * 12 * 64) * sizeof(float); }
template <int KC>
void shape(const char* name, int R, int F, int nseg) {
    SegPtrs s{}; for (int m = 0; m < nseg; ++m) s.p[m] = A + (size_t)m * R * F;
    const double fl = 2.0 * R * (double)(nseg * F) * 192;
    printf("-- %s: R=%d K=%d O=192\n", name, R, nseg * F);
    timeit("gemm_nn_dma<6,KC,2>  (shipped)", fl, gemm_nn_dma_kernel<6, KC, 2>, dim3((R + 127) / 128, 1),
           2 * (size_t)(128 * KC + (KC / 4) * 12 * 64) * sizeof(float), s, nseg, F, R, Bp, 12, bias, C, 192, 192);
    timeit("gemm_nn<6,KC>  (register-staged fallback)", fl, gemm_nn_kernel<6, KC>, dim3((R + 127) / 128, 1), lds3<KC, 4>(), s, nseg, F, R, Bp, 12, bias, C, 192, 192);
}
int main() {
    const size_t R = 291840;
    hipMalloc(&A, 3 * R * 100 * 4); hipMalloc(&Bp, 320 * 192 * 4); hipMalloc(&C, R * 192 * 4); hipMalloc(&bias, 192 * 4);
    hipMemset(A, 0, 3 * R * 100 * 4); hipMemset(Bp, 0, 320 * 192 * 4); hipMemset(bias, 0, 192 * 4);
    shape<16>("layer-1 x-part", (int)R, 64, 3);
    shape<20>("layer-0 x-part", (int)R, 100, 3);
    return 0;
}
