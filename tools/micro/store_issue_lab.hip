// How long does ONE workgroup (4 waves, nothing else on the chip competing for HBM) take to issue the 24 x 16-byte-per-lane
// stores of a 128 x 192 tile, by the shape of the 1 KB an instruction covers?  (gemm_nnq/nnr epilogue: 16 rows x 64 B.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
template <int PAT>
__global__ __launch_bounds__(256) void k(float* __restrict__ C, long long* out, int tiles) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    long long tot = 0;
    for (int t = 0; t < tiles; ++t) {
        float* base = C + (size_t)(blockIdx.x * tiles + t) * 128 * 192;
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int n = 0; n < 24; ++n) {
            size_t off;
            if (PAT == 0) { const int i = n / 3, j = n % 3; off = (size_t)(16 * i + lr) * 192 + 48 * w + 16 * j + 4 * lg; }
            else if (PAT == 1) { const int i = n / 6, j = n % 6; off = (size_t)(32 * w + 8 * i + (lane >> 3)) * 192 + 32 * j + 4 * (lane & 7); }
            else if (PAT == 2) { const int i = n / 3, j = n % 3; off = (size_t)(32 * w + 4 * i + lg) * 192 + 64 * j + 4 * lr; }
            else { off = (size_t)32 * w * 192 + (size_t)n * 256 + 4 * lane; }
            *reinterpret_cast<f32x4*>(base + off) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tot += __builtin_readcyclecounter() - t0;
        for (int s = 0; s < 3000; ++s) v = v * 1.0001f + 0.5f;      // spacing: the tiles of a workgroup are far apart in time
    }
    if (lane == 0) out[blockIdx.x * 4 + w] = tot / tiles;
    if (v[0] == 12345.f) C[0] = v[1];
}
int main() {
    float* C; long long* out;
    const int G = 256, tiles = 8;
    CK(hipMalloc(&C, (size_t)G * tiles * 128 * 192 * 4)); CK(hipMalloc(&out, G * 4 * 8));
    const char* names[4] = {"16 rows x 64 B (wave = 48 columns)", "8 rows x 128 B", "4 rows x 256 B", "1 KB contiguous"};
    for (int g : {1, 256}) for (int p = 0; p < 4; ++p) {
        for (int rep = 0; rep < 2; ++rep) {
            if (p == 0) hipLaunchKernelGGL(k<0>, dim3(g), dim3(256), 0, 0, C, out, tiles);
            if (p == 1) hipLaunchKernelGGL(k<1>, dim3(g), dim3(256), 0, 0, C, out, tiles);
            if (p == 2) hipLaunchKernelGGL(k<2>, dim3(g), dim3(256), 0, 0, C, out, tiles);
            if (p == 3) hipLaunchKernelGGL(k<3>, dim3(g), dim3(256), 0, 0, C, out, tiles);
            CK(hipDeviceSynchronize());
        }
        std::vector<long long> h(g * 4); CK(hipMemcpy(h.data(), out, g * 32, hipMemcpyDeviceToHost));
        double a = 0; for (auto v : h) a += (double)v / h.size();
        printf("%3d workgroups, %-36s: 24 stores + drain = %.0f cycles per wave (%.1f B/clk per workgroup)\n", g, names[p], a, 98304.0 / a);
    }
    return 0;
}
