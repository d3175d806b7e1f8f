// Round 4 lab (MI355X box): one wave per SIMD issuing the MFMA stream of the streamed-weight GEMMs (kernels_decoder.h): per chunk of four
// k-steps NT column tiles x (four 16x16x4 on the 16-node tile + four 4x4x1 on the 4-node remainder), DISTINCT operand registers as in the
// kernels (weights w[i][j], fragments x0[j] / x1[j]; the lab of round 3, chain_lab.hip, reused one operand pair).  Cycles per chunk against
// 32 * 4 NT + 8 * 4 NT, for the orders / chain counts below.  usage: ./nt_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define PIN(v) asm volatile("" : "+v"(v))

// MODE 0: kernel order (j outer, i inner; 16x16x4 pass then 4x4x1 pass, NT*4 remainder chains)
// MODE 1: 16x16x4 only            MODE 2: 4x4x1 only (NT*4 chains)       MODE 3: as 0 with 4 remainder chains in all (rem[j])
// MODE 4: as 0, i outer / j inner in the 4x4x1 pass
// MODE 6: 16x16x4 pass with TWO chains per tile (alternating on j), then the 4x4x1 pass
template <int NT, int MODE>
__global__ __launch_bounds__(256) void nt_kernel(float* __restrict__ buf, long long* __restrict__ out, int n) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float wgt[NT][4], x0[4], x1[4];
    for (int i = 0; i < NT; ++i)
        for (int j = 0; j < 4; ++j) { wgt[i][j] = 0.001f * (lane + 7 * i + j); PIN(wgt[i][j]); }
    for (int j = 0; j < 4; ++j) { x0[j] = 1.f + lane + j; x1[j] = 2.f + lane - j; PIN(x0[j]); PIN(x1[j]); }
    f32x4 acc[NT], alt[NT], rem[NT][4];
    for (int i = 0; i < NT; ++i) {
        acc[i] = alt[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < 4; ++j) rem[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto chunk = [&]() __attribute__((always_inline)) {
        const float (&a0)[4] = x0;
        const float (&a1)[4] = x1;
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    if (MODE == 6 && (j & 1)) alt[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt[i][j], a0[j], alt[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt[i][j], a0[j], acc[i], 0, 0, 0);
                }
        }
        if (MODE != 1) {
            if (MODE == 4) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) rem[i][j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[j], wgt[i][j], rem[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        if (MODE == 3) rem[0][j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[j], wgt[i][j], rem[0][j], 0, 0, 0);
                        else rem[i][j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[j], wgt[i][j], rem[i][j], 0, 0, 0);
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) chunk();
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NT; ++i) {
        s += acc[i][0] + alt[i][1];
        for (int j = 0; j < 4; ++j) s += rem[i][j][0];
    }
    if (s == 12345.f) buf[lane] = s;
    if (lane == 0) out[blockIdx.x * 4 + w] = t1 - t0;
}

template <int NT, int MODE> void run(const char* what, float* buf, long long* out) {
    const int n = 2000, G = 256;
    nt_kernel<NT, MODE><<<G, 256>>>(buf, out, n);
    nt_kernel<NT, MODE><<<G, 256>>>(buf, out, n);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(G * 4);
    CK(hipMemcpy(h.data(), out, sizeof(long long) * G * 4, hipMemcpyDeviceToHost));
    double a = 0;
    for (long long v : h) a += v;
    a /= G * 4.0 * n;
    const int ideal = (MODE == 2 ? 0 : 128 * NT) + (MODE == 1 ? 0 : 32 * NT);
    printf("  NT=%d %-66s %7.1f cycles per chunk (32/8-cycle count: %d)\n", NT, what, a, ideal);
}

int main() {
    float* buf; long long* out;
    CK(hipMalloc(&buf, 4096)); CK(hipMalloc(&out, sizeof(long long) * 256 * 4));
    printf("one wave per SIMD, 256 workgroups x 4 waves, chunk = 4 k-steps x NT tiles x (16x16x4 + 4x4x1):\n");
    run<1, 6>("kernel order, two 16x16x4 chains (as shipped for one tile)", buf, out);
    run<1, 0>("kernel order, one 16x16x4 chain", buf, out);
    run<2, 0>("kernel order", buf, out);
    run<3, 0>("kernel order", buf, out);
    run<2, 1>("16x16x4 only", buf, out);
    run<3, 1>("16x16x4 only", buf, out);
    run<2, 2>("4x4x1 only, 4 NT chains", buf, out);
    run<3, 2>("4x4x1 only, 4 NT chains", buf, out);
    run<2, 3>("kernel order, 4 remainder chains in all", buf, out);
    run<3, 3>("kernel order, 4 remainder chains in all", buf, out);
    run<2, 4>("4x4x1 pass tile by tile", buf, out);
    run<3, 4>("4x4x1 pass tile by tile", buf, out);
    run<2, 6>("two 16x16x4 chains per tile", buf, out);
    run<3, 6>("two 16x16x4 chains per tile", buf, out);
    return 0;
}
