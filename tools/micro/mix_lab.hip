// Do the global stores of one wave slow down the fp32 MFMAs of ANOTHER wave on the same SIMD?  (round 3: the NN GEMM loses
// exactly its store time, as if nothing overlapped.)  One workgroup per CU, 8 waves: waves 0-3 issue NM independent MFMAs
// from registers, waves 4-7 (same SIMDs) run `mode`: 0 idle, 1 16-byte stores (the GEMM epilogue pattern), 2 LDS-DMA loads,
// 3 plain 16-byte loads.  Prints the cycles of the MFMA waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void mix_kernel(float* __restrict__ buf, long long* __restrict__ out, int nm, int mode, int nmem) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* mine = buf + (size_t)blockIdx.x * (1 << 20);            // 4 MB per workgroup
    if (w < 4) {
        f32x4 acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float a = lane * 0.001f, b = 1.f + lane;
        const long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < nm; ++it)
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        const long long t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 12; ++i) s += acc[i][0];
        if (s == 12345.f) mine[lane] = s;
        if (lane == 0) out[blockIdx.x * 4 + w] = t1 - t0;
    } else if (mode == 1) {
        f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
        for (int it = 0; it < nmem; ++it) {
            // 16 rows x 64 B, row stride 768 B, 24 per "tile"
            const int i = it % 8, j = (it / 8) % 3, t = it / 24;
            *reinterpret_cast<f32x4*>(mine + (size_t)(t % 40) * 24576 + (size_t)(16 * i + (lane & 15)) * 192 + 48 * (w - 4) + 16 * j + 4 * (lane >> 4)) = v;
        }
    } else if (mode == 2) {
        for (int it = 0; it < nmem; ++it)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mine + (size_t)(it % 4000) * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(sm + (w - 4) * 1024 + (it & 3) * 256), 16, 0, 0);
    } else if (mode == 3) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < nmem; ++it) s += *reinterpret_cast<const f32x4*>(mine + (size_t)(it % 4000) * 256 + lane * 4);
        if (s[0] == 12345.f) mine[lane] = s[1];
    } else if (mode == 4 || mode == 5 || mode == 6) {
        // VALU partner: nmem x 16 independent fp32 FMAs (mode 4), integer adds (mode 5), v_exp_f32 (mode 6)
        float a[16]; int b[16];
        for (int i = 0; i < 16; ++i) { a[i] = lane + i; b[i] = lane * i; }
        const long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < nmem; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (mode == 4) a[i] = a[i] * 1.0001f + 0.5f;
                else if (mode == 5) b[i] = b[i] + it;
                else a[i] = __builtin_amdgcn_exp2f(a[i] * 0.001f);
            }
        }
        const long long t1 = __builtin_readcyclecounter();
        float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + b[i];
        if (s == 12345.f) mine[lane] = s;
        if (lane == 0) out[1024 + blockIdx.x * 4 + (w - 4)] = t1 - t0;
    }
}

int main() {
    float* buf; long long* out;
    CK(hipMalloc(&buf, (size_t)256 * (1 << 20) * 4)); CK(hipMalloc(&out, 2 * 256 * 4 * 8)); CK(hipMemset(out, 0, 2 * 256 * 4 * 8));
    CK(hipMemset(buf, 0, (size_t)256 * (1 << 20) * 4));
    const int nm = 2000;                                            // 24 000 MFMAs per wave = 768 k cycles at the issue rate
    const char* names[4] = {"idle", "16-byte stores (GEMM epilogue pattern)", "LDS-DMA loads", "plain 16-byte loads"};
    for (int mode = 0; mode < 4; ++mode)
        for (int nmem : {2000, 8000}) {
            if (mode == 0 && nmem != 2000) continue;
            hipLaunchKernelGGL(mix_kernel, dim3(256), dim3(512), 32768, 0, buf, out, nm, mode, nmem);
            CK(hipDeviceSynchronize());
            std::vector<long long> h(1024); CK(hipMemcpy(h.data(), out, 8192, hipMemcpyDeviceToHost));
            double avg = 0; for (auto v : h) avg += v / 1024.0;
            printf("partner waves: %-40s x %5d per wave -> MFMA waves %.0f cycles for %d MFMAs = %.1f cycles/MFMA (32.0 = issue rate)\n", names[mode], nmem, avg, nm * 12, avg / (nm * 12));
        }
    // VALU partners: how many cycles does a VALU instruction take beside an fp32 MFMA stream, and what does it cost the MFMAs?
    const char* vn[3] = {"fp32 FMA", "int add", "v_exp_f32"};
    for (int mode = 4; mode <= 6; ++mode)
        for (int alone = 0; alone < 2; ++alone) {
            const int nv = 20000;
            hipLaunchKernelGGL(mix_kernel, dim3(256), dim3(512), 32768, 0, buf, out, alone ? 1 : nm, mode, nv);
            CK(hipDeviceSynchronize());
            std::vector<long long> h(2048); CK(hipMemcpy(h.data(), out, 16384, hipMemcpyDeviceToHost));
            double am = 0, av = 0; for (int i = 0; i < 1024; ++i) { am += h[i] / 1024.0; av += h[1024 + i] / 1024.0; }
            printf("partner waves: %d x 16 %-9s %s: VALU waves %.1f cycles per instruction; MFMA waves %.1f cycles/MFMA\n", nv, vn[mode - 4],
                   alone ? "ALONE (no MFMA stream) " : "beside the MFMA stream", av / (nv * 16.0), alone ? 0.0 : am / (nm * 12.0));
        }
    return 0;
}
