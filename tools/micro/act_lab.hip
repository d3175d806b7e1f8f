// Round 4 lab (MI355X box): (1) lane maps of gfx950's v_permlane32_swap_b32 / v_permlane16_swap_b32 and the remainder
// reduce-scatter built on them (common.h rem4_reduce); (2) what sigmoid / tanh cost on one SIMD: the shipped v_exp_f32 +
// v_rcp_f32 forms against polynomial / rational forms on v_pk_fma_f32, alone and beside a wave that streams fp32 MFMAs on the
// same SIMD (for fp32 the SIMD is one issue resource: DESIGN.md 4.1).  usage: ./act_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void permlane32_swap(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void permlane16_swap(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

__global__ void perm_kernel(float* out) {
    const int l = threadIdx.x;
    float a = 100.f + l, b = 200.f + l;
    permlane32_swap(a, b);
    out[l] = a; out[64 + l] = b;
    float c = 100.f + l, d = 200.f + l;
    permlane16_swap(c, d);
    out[128 + l] = c; out[192 + l] = d;
    // rem4_reduce: register r of lane (lr, lg) = partial of group lg for (node 16 + r, col lr): value 1000*r + 10*lr + (lg+1)*0.001
    const int lr = l & 15, lg = l >> 4;
    float t0 = 0 * 1000.f + lr * 10.f + (lg + 1), t1 = 1000.f + lr * 10.f + (lg + 1), t2 = 2000.f + lr * 10.f + (lg + 1), t3 = 3000.f + lr * 10.f + (lg + 1);
    permlane32_swap(t0, t2);
    permlane32_swap(t1, t3);
    float s02 = t0 + t2, s13 = t1 + t3;
    permlane16_swap(s02, s13);
    out[256 + l] = s02 + s13;      // expected: 4 * (1000*lg + 10*lr) + 10
}

// ---- activation cost: NV float4 activations per iteration on every lane; optional MFMA partner wave on the same SIMD
__device__ __forceinline__ f32x4 sig_exp(f32x4 x) {        // shipped form: 2 pk_mul, 4 v_exp, 2 pk_add, 4 v_rcp
    const f32x4 z = x * -1.44269504088896340736f;
    f32x4 d = {__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1]), __builtin_amdgcn_exp2f(z[2]), __builtin_amdgcn_exp2f(z[3])};
    d = d + 1.0f;
    return (f32x4){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
}
// rational tanh (the widely used float form: clamp to +-7.9, odd degree-13 numerator / even degree-6 denominator, ~1e-7 abs) and
// sigmoid(x) = 0.5 + 0.5 tanh(x/2): per float4 4 v_med3 (clamp) + 2 pk_mul (x^2) + 2*6 pk_fma (numerator) + 2 pk_mul + 2*3 pk_fma
// (denominator) + 4 v_rcp + 2 pk_mul + 2 pk_fma = 32 instructions, 4 of them quarter rate -- no v_exp
__device__ __forceinline__ f32x4 tanh_rat(f32x4 x) {
    f32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_fmed3f(x[i], -7.90531110763549805f, 7.90531110763549805f);
    const f32x4 x2 = c * c;
    f32x4 p = x2 * -2.76076847742355e-16f + 2.00018790482477e-13f;
    p = p * x2 + -8.60467152213735e-11f;
    p = p * x2 + 5.12229709037114e-08f;
    p = p * x2 + 1.48572235717979e-05f;
    p = p * x2 + 6.37261928875436e-04f;
    p = p * x2 + 4.89352455891786e-03f;
    p = p * c;
    f32x4 q = x2 * 1.19825839466702e-06f + 1.18534705686654e-04f;
    q = q * x2 + 2.26843463243900e-03f;
    q = q * x2 + 4.89352518554385e-03f;
    return p * (f32x4){__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1]), __builtin_amdgcn_rcpf(q[2]), __builtin_amdgcn_rcpf(q[3])};
}
__device__ __forceinline__ f32x4 sig_rat(f32x4 x) { return tanh_rat(x * 0.5f) * 0.5f + 0.5f; }

template <int KIND, int PARTNER>   // KIND 0: exp+rcp sigmoid, 1: rational sigmoid, 2: nothing (partner alone)
__global__ __launch_bounds__(512) void act_kernel(float* __restrict__ buf, long long* __restrict__ out, int iters, float* __restrict__ err) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __syncthreads();
    if (w < 4) {
        f32x4 x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = (f32x4){0.01f * lane + i, -0.02f * lane + i, 0.3f + i, -1.f - i};
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        const long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KIND == 0) s += sig_exp(x[i]);
                if (KIND == 1) s += sig_rat(x[i]);
                x[i] += s * 1e-9f;          // dependent: nothing hoists
            }
        }
        const long long t1 = __builtin_readcyclecounter();
        if (s[0] + s[1] + s[2] + s[3] == 12345.f) buf[lane] = s[0];
        if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
        if (err != nullptr && blockIdx.x == 0 && w == 0) {          // accuracy of the two forms on a sweep (lane-parallel)
            float worst0 = 0.f, worst1 = 0.f;
            for (int k = 0; k < 4096; ++k) {
                const float v = -20.f + 40.f * (k * 64 + lane) / (4096.f * 64.f);
                const f32x4 vv = {v, v, v, v};
                const float ref = 1.f / (1.f + expf(-v));
                worst0 = fmaxf(worst0, fabsf(sig_exp(vv)[0] - ref));
                worst1 = fmaxf(worst1, fabsf(sig_rat(vv)[0] - ref));
            }
            err[lane] = worst0; err[64 + lane] = worst1;
        }
    } else if (PARTNER) {
        const float a = lane * 0.001f, b = 1.f + lane;
        f32x4 p = {0.f, 0.f, 0.f, 0.f}, q = p;
        const long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters * PARTNER; ++it) { p = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, p, 0, 0, 0); q = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q, 0, 0, 0); }
        const long long t1 = __builtin_readcyclecounter();
        if (p[0] + q[0] == 12345.f) buf[lane] = p[0];
        if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
    } else if (lane == 0) out[blockIdx.x * 8 + w] = 0;
}

template <int KIND, int PARTNER> void run_act(const char* name, float* buf, long long* out, float* err) {
    const int iters = 2000, G = 256;
    act_kernel<KIND, PARTNER><<<G, 512>>>(buf, out, iters, nullptr);
    act_kernel<KIND, PARTNER><<<G, 512>>>(buf, out, iters, err);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(G * 8);
    CK(hipMemcpy(h.data(), out, sizeof(long long) * G * 8, hipMemcpyDeviceToHost));
    double a = 0, b = 0;
    for (int g = 0; g < G; ++g) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += h[g * 8 + w];
    a /= G * 4.0 * iters * 4; b /= G * 4.0 * iters * (PARTNER ? 2.0 * PARTNER : 1.0);
    printf("  %-44s activation wave: %7.1f cycles per float4 sigmoid%s", name, a, KIND == 2 ? " (none)" : "");
    if (PARTNER) printf("   MFMA partner wave: %6.1f cycles per 16x16x4 MFMA (%d per activation iteration)", b, 2 * PARTNER);
    printf("\n");
}

int main() {
    float* buf; long long* out; float* err;
    CK(hipMalloc(&buf, 4096)); CK(hipMalloc(&out, sizeof(long long) * 256 * 8)); CK(hipMalloc(&err, 4 * 128));
    perm_kernel<<<1, 64>>>(buf);
    CK(hipDeviceSynchronize());
    float h[320];
    CK(hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost));
    printf("v_permlane32_swap_b32 a, b  (a = 100 + lane, b = 200 + lane)\n  a:");
    for (int l = 0; l < 64; l += 8) printf(" %g", h[l]);
    printf("\n  b:");
    for (int l = 0; l < 64; l += 8) printf(" %g", h[64 + l]);
    printf("\nv_permlane16_swap_b32 c, d\n  c:");
    for (int l = 0; l < 64; l += 8) printf(" %g", h[128 + l]);
    printf("\n  d:");
    for (int l = 0; l < 64; l += 8) printf(" %g", h[192 + l]);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const bool hi = l >= 32, odd = (l >> 4) & 1;
        bad += h[l] != (hi ? 200.f + (l - 32) : 100.f + l);            // a.hi <- b.lo
        bad += h[64 + l] != (hi ? 200.f + l : 100.f + (l + 32));       // b.lo <- a.hi
        bad += h[128 + l] != (odd ? 200.f + (l - 16) : 100.f + l);     // c.odd rows <- d.even rows
        bad += h[192 + l] != (odd ? 200.f + l : 100.f + (l + 16));     // d.even rows <- c.odd rows
        bad += h[256 + l] != 4.f * (1000.f * (l >> 4) + 10.f * (l & 15)) + 10.f;
    }
    printf("\nlane maps + rem4_reduce: %s (%d mismatches)\n", bad == 0 ? "as documented" : "DIFFERENT", bad);

    printf("sigmoid of a float4 per lane, 4 SIMDs x 256 workgroups (cycles from s_memtime; one activation wave per SIMD):\n");
    run_act<0, 0>("v_exp_f32 + v_rcp_f32 (shipped), alone", buf, out, err);
    float e[128];
    CK(hipMemcpy(e, err, sizeof(e), hipMemcpyDeviceToHost));
    float w0 = 0, w1 = 0;
    for (int l = 0; l < 64; ++l) { w0 = fmaxf(w0, e[l]); w1 = fmaxf(w1, e[64 + l]); }
    run_act<1, 0>("rational on v_pk_fma_f32 + 1 v_rcp, alone", buf, out, err);
    run_act<0, 4>("v_exp + v_rcp beside an MFMA wave", buf, out, err);
    run_act<1, 4>("rational beside an MFMA wave", buf, out, err);
    run_act<2, 4>("MFMA wave alone", buf, out, err);
    printf("max |err| vs 1/(1+expf(-x)) on [-20, 20]: exp+rcp %.2e, rational %.2e\n", w0, w1);
    return 0;
}
