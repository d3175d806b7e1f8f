// Two waves on one SIMD, both issuing fp32 MFMAs (round 3, the recurrent kernels: role A's r-gate GEMM is a DEPENDENT chain on
// one accumulator, role B's u-gate GEMM runs beside it).  How long does wave A (priority PRIO) take for NA MFMAs spread over CH
// accumulator chains, alone and beside a wave B that streams independent MFMAs?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int CH, int PRIO, int BCH>
__global__ __launch_bounds__(512) void chain_kernel(float* __restrict__ buf, long long* __restrict__ out, int na, int nb) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float a = lane * 0.001f, b = 1.f + lane;
    __syncthreads();
    if (w < 4) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        f32x4 acc[CH];
        for (int i = 0; i < CH; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < na / CH; ++it)
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        const long long t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < CH; ++i) s += acc[i][0];
        if (s == 12345.f) buf[lane] = s;
        if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
    } else {
        f32x4 acc[BCH];
        for (int i = 0; i < BCH; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < nb / BCH; ++it)
#pragma unroll
            for (int i = 0; i < BCH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        const long long t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < BCH; ++i) s += acc[i][0];
        if (s == 12345.f) buf[lane] = s;
        if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
    }
}

// One wave per SIMD: groups of G16 16x16x4 MFMAs (two chains) followed by G4 4x4x1 MFMAs (four chains): the k-loop of the recurrent
// kernels' GEMM over 20 nodes (16-node tile + 4-node remainder).  Cycles per group against 32 * G16 + 8 * G4.
template <int G16, int G4>
__global__ __launch_bounds__(256) void shape_kernel(float* __restrict__ buf, long long* __restrict__ out, int n) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float a = lane * 0.001f, b = 1.f + lane;
    f32x4 p = {0.f, 0.f, 0.f, 0.f}, q = p, r0 = p, r1 = p, r2 = p, r3 = p;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if constexpr (G16 == 1 && G4 == 1) {          // strict alternation (the decoder kernels' order), two chains of each shape
                p = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, p, 0, 0, 0); r0 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, r0, 0, 0, 0);
                q = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q, 0, 0, 0); r1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, r1, 0, 0, 0);
            } else {
#pragma unroll
            for (int i = 0; i < G16 / 2; ++i) { p = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, p, 0, 0, 0); q = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q, 0, 0, 0); }
#pragma unroll
            for (int i = 0; i < G4 / 4; ++i) {
                r0 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, r0, 0, 0, 0); r1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, r1, 0, 0, 0);
                r2 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, r2, 0, 0, 0); r3 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, r3, 0, 0, 0);
            }
            if (G4 % 4 >= 1) r0 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, r0, 0, 0, 0);
            if (G4 % 4 >= 2) r1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, r1, 0, 0, 0);
            if (G4 % 4 >= 3) r2 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, r2, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = p[0] + q[0] + r0[0] + r1[0] + r2[0] + r3[0];
    if (s == 12345.f) buf[lane] = s;
    if (lane == 0) out[blockIdx.x * 4 + w] = t1 - t0;
}
template <int G16, int G4> void run_shape(float* buf, long long* out) {
    const int n = 2000;
    hipLaunchKernelGGL((shape_kernel<G16, G4>), dim3(256), dim3(256), 0, 0, buf, out, n);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(1024); CK(hipMemcpy(h.data(), out, 1024 * 8, hipMemcpyDeviceToHost));
    double t = 0; for (auto v : h) t += v / 1024.0;
    if (G16 == 1 && G4 == 1) printf("strict alternation 16x16x4, 4x4x1, ... (2 + 2 per group): %.1f cycles per group (80)\n", t / n / 4);
    else printf("groups of %2d 16x16x4 + %2d 4x4x1: %.1f cycles per group (32 x %d + 8 x %d = %d)\n", G16, G4, t / n / 4, G16, G4, 32 * G16 + 8 * G4);
}

// Wave A streams 16x16x4 MFMAs (two chains), wave B on the same SIMD streams 4x4x1 MFMAs (four chains): is the cost of changing
// shapes a property of the pipe (then the two streams disturb each other) or of one wave's instruction order?
__global__ __launch_bounds__(512) void two_shape_kernel(float* __restrict__ buf, long long* __restrict__ out, int na, int nb) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float a = lane * 0.001f, b = 1.f + lane;
    __syncthreads();
    f32x4 p = {0.f, 0.f, 0.f, 0.f}, q = p, r2 = p, r3 = p;
    const long long t0 = __builtin_readcyclecounter();
    if (w < 4) {
        for (int it = 0; it < na / 4; ++it) {
            p = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, p, 0, 0, 0); q = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, p, 0, 0, 0); q = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q, 0, 0, 0);
        }
    } else {
        for (int it = 0; it < nb / 4; ++it) {
            p = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, p, 0, 0, 0); q = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, q, 0, 0, 0);
            r2 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, r2, 0, 0, 0); r3 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, r3, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const float s = p[0] + q[0] + r2[0] + r3[0];
    if (s == 12345.f) buf[lane] = s;
    if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
}
void run_two_shape(float* buf, long long* out, int na, int nb) {
    hipLaunchKernelGGL(two_shape_kernel, dim3(256), dim3(512), 0, 0, buf, out, na, nb);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(2048); CK(hipMemcpy(h.data(), out, 2048 * 8, hipMemcpyDeviceToHost));
    double ta = 0, tb = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? ta : tb) += h[b * 8 + w] / 1024.0;
    printf("wave A: %d 16x16x4 (%d cycles at 32) in %.0f cycles | wave B: %d 4x4x1 (%d cycles at 8) in %.0f cycles | one after the other: %d\n",
           na, na * 32, ta, nb, nb * 8, tb, na * 32 + nb * 8);
}

// VALU beside the matrix pipe.  SAME: one wave per SIMD issues NV v_pk_fma_f32 behind every 16x16x4 MFMA (two chains).  CROSS: wave
// A streams MFMAs, wave B on the same SIMD streams v_pk_fma_f32.  (Would the 4-node remainder of the recurrent GEMMs be cheaper on the
// VALU than as 4x4x1 MFMAs?)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NV, bool CROSS>
__global__ __launch_bounds__(512) void valu_kernel(float* __restrict__ buf, long long* __restrict__ out, int n) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float a = lane * 0.001f, b = 1.f + lane;
    f32x4 p = {0.f, 0.f, 0.f, 0.f}, q = p;
    f32x2 v[8], x = {a, b}, y = {b, a};
    for (int i = 0; i < 8; ++i) v[i] = (f32x2){(float)i, a};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (!CROSS || w < 4) {
        if (CROSS || w < 4)
        for (int it = 0; it < n; ++it) {
            p = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, p, 0, 0, 0);
            if (!CROSS) {
#pragma unroll
                for (int i = 0; i < NV; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(x), "v"(y));
            }
            q = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q, 0, 0, 0);
            if (!CROSS) {
#pragma unroll
                for (int i = 0; i < NV; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(x), "v"(y));
            }
        }
    } else {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(x), "v"(y));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = p[0] + q[0];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    if (s == 12345.f) buf[lane] = s;
    if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
}
template <int NV, bool CROSS> void run_valu(float* buf, long long* out) {
    const int n = 2000;
    CK(hipMemset(out, 0, 2048 * 8));
    hipLaunchKernelGGL((valu_kernel<NV, CROSS>), dim3(256), dim3(512), 0, 0, buf, out, n);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(2048); CK(hipMemcpy(h.data(), out, 2048 * 8, hipMemcpyDeviceToHost));
    double ta = 0, tb = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? ta : tb) += h[b * 8 + w] / 1024.0;
    if (!CROSS) printf("same wave: %d v_pk_fma_f32 behind every 16x16x4 MFMA: %.1f cycles per MFMA (+ its VALU)\n", NV, ta / (2.0 * n));
    else printf("cross wave: A %.1f cycles per MFMA | B %.2f cycles per v_pk_fma_f32 (%d per A-MFMA slot)\n", ta / (2.0 * n), tb / (2.0 * n * NV), NV);
}

template <int CH, int PRIO, int BCH> void run(float* buf, long long* out, int na, int nb, const char* what) {
    hipLaunchKernelGGL((chain_kernel<CH, PRIO, BCH>), dim3(256), dim3(512), 0, 0, buf, out, na, nb);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(2048); CK(hipMemcpy(h.data(), out, 2048 * 8, hipMemcpyDeviceToHost));
    double ta = 0, tb = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? ta : tb) += h[b * 8 + w] / 1024.0;
    printf("A: %d chains%s | B: %-28s -> A %.1f cycles/MFMA (%d MFMAs)%s\n", CH, PRIO ? ", priority 3" : "            ", what, ta / na, na,
           nb ? (std::string(";  B ") + std::to_string(tb / nb).substr(0, 5) + " cycles/MFMA").c_str() : "");
}

int main() {
    float* buf; long long* out;
    CK(hipMalloc(&buf, 1 << 20)); CK(hipMalloc(&out, 2048 * 8));
    const int na = 4800;
    run<1, 0, 12>(buf, out, na, 0, "idle");            run<2, 0, 12>(buf, out, na, 0, "idle");            run<4, 0, 12>(buf, out, na, 0, "idle");
    run<1, 0, 12>(buf, out, na, 9600, "12 chains, 2x the MFMAs");  run<2, 0, 12>(buf, out, na, 9600, "12 chains, 2x the MFMAs");  run<4, 0, 12>(buf, out, na, 9600, "12 chains, 2x the MFMAs");
    run<1, 1, 12>(buf, out, na, 9600, "12 chains, 2x the MFMAs");  run<2, 1, 12>(buf, out, na, 9600, "12 chains, 2x the MFMAs");  run<4, 1, 12>(buf, out, na, 9600, "12 chains, 2x the MFMAs");
    run<1, 0, 1>(buf, out, na, 4800, "1 chain, as many MFMAs");    run<1, 1, 1>(buf, out, na, 4800, "1 chain, as many MFMAs");
    run<2, 1, 1>(buf, out, na, 4800, "1 chain, as many MFMAs");    run<2, 1, 2>(buf, out, na, 4800, "2 chains, as many MFMAs");
    run_valu<0, false>(buf, out); run_valu<2, false>(buf, out); run_valu<4, false>(buf, out); run_valu<6, false>(buf, out); run_valu<8, false>(buf, out); run_valu<12, false>(buf, out);
    run_valu<4, true>(buf, out); run_valu<8, true>(buf, out); run_valu<16, true>(buf, out);
    run_two_shape(buf, out, 4800, 0); run_two_shape(buf, out, 0, 4800); run_two_shape(buf, out, 4800, 4800); run_two_shape(buf, out, 4800, 19200);
    run_shape<4, 0>(buf, out); run_shape<0, 4>(buf, out); run_shape<0, 16>(buf, out); run_shape<4, 4>(buf, out); run_shape<8, 8>(buf, out); run_shape<16, 16>(buf, out); run_shape<4, 1>(buf, out); run_shape<1, 1>(buf, out); run_shape<2, 2>(buf, out); run_shape<2, 1>(buf, out); run_shape<4, 2>(buf, out); run_shape<4, 3>(buf, out); run_shape<48, 48>(buf, out);
    return 0;
}
