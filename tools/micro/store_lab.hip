// Store-pattern probe (round 3): how fast does a CU retire 16-byte-per-lane global stores, by the shape of the 1 KB a wave
// instruction covers?  The NN GEMM epilogue stores 16 rows x 64 B per instruction (row stride = ldc * 4 = 768 B).
//   ./store_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// every workgroup (4 waves) writes `tiles` tiles of 128 rows x 192 floats (96 KB), tile t of block b at rows (b * tiles + t) * 128.
// PAT 0: wave = 48-column group, instruction = 16 rows x 64 B        (the gemm_nnq epilogue)
// PAT 1: wave = 32-row group,   instruction = 16 rows x 64 B, the two halves of a 128-B line back to back
// PAT 2: wave = 32-row group,   instruction = 8 rows x 128 B (full lines)
// PAT 3: wave = 32-row group,   instruction = 4 rows x 256 B
// PAT 4: wave = 32-row group,   instruction = 1 KB contiguous = 1.33 rows
template <int PAT>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ C, int tiles, int spin) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    for (int t = 0; t < tiles; ++t) {
        float* base = C + (size_t)(blockIdx.x * tiles + t) * 128 * 192;
        // some arithmetic between the tiles (stands for the MFMA chunks): spin dependent FMAs
        for (int s = 0; s < spin; ++s) v = v * 1.0001f + 0.5f;
#pragma unroll
        for (int n = 0; n < 24; ++n) {
            size_t off;
            if (PAT == 0) { const int i = n / 3, j = n % 3; off = (size_t)(16 * i + lr) * 192 + 48 * w + 16 * j + 4 * lg; }
            else if (PAT == 1) { const int i = n / 12, j = n % 12; off = (size_t)(32 * w + 16 * i + lr) * 192 + 16 * j + 4 * lg; }
            else if (PAT == 2) { const int i = n / 6, j = n % 6; off = (size_t)(32 * w + 8 * i + (lane >> 3)) * 192 + 32 * j + 4 * (lane & 7); }
            else if (PAT == 3) { const int i = n / 3, j = n % 3; off = (size_t)(32 * w + 4 * i + lg) * 192 + 64 * j + 4 * lr; }
            else { off = (size_t)32 * w * 192 + (size_t)n * 256 + 4 * lane; }
            *reinterpret_cast<f32x4*>(base + off) = v;
        }
    }
}

template <int PAT> float run(float* C, int G, int tiles, int spin) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < 7; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(store_kernel<PAT>, dim3(G), dim3(256), 0, 0, C, tiles, spin);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main() {
    const int G = 512, tiles = 5;                          // 512 x 5 x 96 KB = 252 MB (the NN GEMM writes 224 MB)
    float* C; CK(hipMalloc(&C, (size_t)G * tiles * 128 * 192 * 4));
    const char* names[5] = {"16 rows x 64 B, wave = 48 columns (nnq epilogue)", "16 rows x 64 B, wave = 32 rows, line halves adjacent",
                            "8 rows x 128 B", "4 rows x 256 B", "1 KB contiguous"};
    for (int spin = 0; spin <= 4000; spin += 4000) {
        printf("spin %d FMAs between tiles:\n", spin);
        float t[5] = {run<0>(C, G, tiles, spin), run<1>(C, G, tiles, spin), run<2>(C, G, tiles, spin), run<3>(C, G, tiles, spin), run<4>(C, G, tiles, spin)};
        for (int p = 0; p < 5; ++p) {
            const double bytes = (double)G * tiles * 128 * 192 * 4, instr_per_cu = (double)G / 256 * 4 * tiles * 24;
            printf("  %-52s %.4f ms  %.2f TB/s  %.0f cycles per store instruction per CU (2.1 GHz)\n", names[p], t[p], bytes / t[p] / 1e9, t[p] * 1e-3 * 2.1e9 / instr_per_cu);
        }
    }
    return 0;
}
