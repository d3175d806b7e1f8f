// Prints the operand/result lane layout of v_mfma_f32_4x4x1_16B_f32 (used by the remainder-node path).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), (float)(1000 * (l + 1)), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); float h[256];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) { long v = (long)(h[l * 4 + r] + 0.5f); printf("  a=lane%2ld b=lane%2ld", (v / 1000) ? 0 : 0, 0L); (void)v; }
        printf("   raw %.0f %.0f %.0f %.0f\n", h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
