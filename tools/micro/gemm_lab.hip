// GEMM lab (round 3): stand-alone A/B harness for the hoisted NN GEMM of the DCGRU layer at the cfg2 shapes.
// Baseline = the round-2 product kernel (gemm_nn_dma_kernel); candidate = the round-3 design (gemm_nnq_kernel, kernels_gemm_q.h).
// Random operands (zero-filled ones clock ~15 % higher: cdna_hip_programming.md 5.4 rule 25), interleaved rounds, median.
//   make -C tools/micro gemm_lab && ./gemm_lab [rounds]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "../../eeg_gnn_ssl_amd/csrc/kernels_gemm.h"
#include "gemm_lab_kernels.h"
using namespace eeg;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static void fill_rand(float* d, size_t n, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
}

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };

static void run_variants(std::vector<Variant>& vs, int rounds, double flops) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& v : vs) { v.launch(); }                       // warm
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0));
            v.launch(); v.launch(); v.launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / 3);
        }
    CK(hipGetLastError());
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("  %-44s med %.4f ms  min %.4f ms  %6.1f TF/s (med)  frac %.3f\n", v.name.c_str(), med, mn, flops / med / 1e9, flops / med / 1e9 / 157.3);
    }
}

// ---- host reference + packs ------------------------------------------------------------------
// logical weight W[k = seg*F + f][o]; old pack: fragment order; new pack: nnq order (kernels_gemm_q.h)
static void pack_old(const std::vector<float>& W, int K, int O, std::vector<float>& out) {
    const int nct = O / 16; out.assign((size_t)K * O, 0.f);
    for (size_t e = 0; e < out.size(); ++e) {
        const int lane = e & 63, ct = (e >> 6) % nct, ks = (e >> 6) / nct;
        out[e] = W[(size_t)(4 * ks + (lane >> 4)) * O + 16 * ct + (lane & 15)];
    }
}
static void pack_q(const std::vector<float>& W, int nseg, int F, int O, std::vector<float>& out) {
    const NnqOrder ko = make_nnq_order(nseg, F);
    const int nct = O / 16; out.assign((size_t)ko.nch * nct * 256, 0.f);
    for (size_t e = 0; e < out.size(); ++e) {
        const int s = e & 3, lane = (e >> 2) & 63, ct = (e >> 8) % nct, c = (e >> 8) / nct;
        const int k = nnq_k_of(ko, c, lane >> 4, s);
        out[e] = k < 0 ? 0.f : W[(size_t)k * O + 16 * ct + (lane & 15)];
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 15;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    const int R = 291840, O = 192;
    float *A, *Bold, *Bq, *C, *C2, *bias;
    CK(hipMalloc(&A, (size_t)5 * R * 100 * 4)); CK(hipMalloc(&Bold, 512 * 192 * 4)); CK(hipMalloc(&Bq, 512 * 192 * 4));
    CK(hipMalloc(&C, (size_t)R * O * 4)); CK(hipMalloc(&C2, (size_t)R * O * 4)); CK(hipMalloc(&bias, 192 * 4));
    fill_rand(A, (size_t)5 * R * 100, 1); fill_rand(bias, 192, 3);

    struct Shape { const char* name; int nseg, F; int btT, btB, btN; };
    const Shape shapes[] = {{"layer-1 x-part / dX (K=192)", 3, 64, 0, 0, 0}, {"layer-0 x-part (K=300, batch-major rows)", 3, 100, 60, 256, 19},
                            {"K=300 time-major", 3, 100, 0, 0, 0}, {"K=192 batch-major", 3, 64, 60, 256, 19}, {"K=320 (5 x 64)", 5, 64, 0, 0, 0}};
    for (int si = 0; si < 5; ++si) {
        const Shape sh = shapes[si];
        if (only >= 0 && only != si) continue;
        const int K = sh.nseg * sh.F;
        std::vector<float> W((size_t)K * O), po, pq;
        { unsigned s = 77 + si; for (auto& w : W) { s = s * 1664525u + 1013904223u; w = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; } }
        pack_old(W, K, O, po); pack_q(W, sh.nseg, sh.F, O, pq);
        CK(hipMemcpy(Bold, po.data(), po.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(Bq, pq.data(), pq.size() * 4, hipMemcpyHostToDevice));
        SegPtrs segs{}; for (int m = 0; m < sh.nseg; ++m) segs.p[m] = A + (size_t)m * R * sh.F;
        const double fl = 2.0 * R * (double)K * O;
        printf("== %s: R=%d K=%d O=%d\n", sh.name, R, K, O);

        // ---- correctness of the candidate against the baseline kernel (full size, every element) + a host spot check
        auto base = [&](float* out) {
            if (sh.F % 16 == 0) {
                const size_t lds = 2 * (size_t)(128 * 16 + 4 * 12 * 64) * 4;
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nn_dma_kernel<6, 16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((gemm_nn_dma_kernel<6, 16, 2>), dim3((R + 127) / 128, 1), dim3(256), lds, 0, segs, sh.nseg, sh.F, R, Bold, 12, bias, out, O, O, sh.btT, sh.btB, sh.btN);
            } else {
                const size_t lds = 2 * (size_t)(128 * 20 + 5 * 12 * 64) * 4;
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nn_dma_kernel<6, 20, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((gemm_nn_dma_kernel<6, 20, 2>), dim3((R + 127) / 128, 1), dim3(256), lds, 0, segs, sh.nseg, sh.F, R, Bold, 12, bias, out, O, O, sh.btT, sh.btB, sh.btN);
            }
        };
        auto cand = [&](auto kern, int ns, int G, float* out, int Rr, int flags = 0) {
            const size_t lds = (size_t)ns * kNnqStageFloats * 4;
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(G, 1), dim3(256), lds, 0, segs, sh.nseg, sh.F, Rr, Bq, 12, bias, out, O, O, sh.btT, sh.btB, sh.btN, flags, (long long*)nullptr);
        };
        {
            CK(hipMemset(C, 0, (size_t)R * O * 4)); CK(hipMemset(C2, 0xff, (size_t)R * O * 4));
            base(C); cand(gemm_nnq_kernel<4, 0>, 4, 512, C2, R);
            CK(hipDeviceSynchronize());
            std::vector<float> h1((size_t)R * O), h2((size_t)R * O);
            CK(hipMemcpy(h1.data(), C, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost));
            double maxd = 0, maxv = 0; size_t bad = 0;
            for (size_t i = 0; i < h1.size(); ++i) { const double d = std::fabs((double)h1[i] - h2[i]); if (!(d <= 1e-3)) ++bad; if (d > maxd) maxd = d; if (std::fabs(h1[i]) > maxv) maxv = std::fabs(h1[i]); }
            printf("  nnq<4> vs baseline, full size: max |diff| %.3e (max |value| %.2f), elements off by > 1e-3: %zu\n", maxd, maxv, bad);
            // ragged: R' not a multiple of 16 and G that does not divide the row tiles
            const int Rr = sh.btT > 0 ? R : 100003;
            CK(hipMemset(C2, 0xff, (size_t)R * O * 4));
            cand(gemm_nnq_kernel<3, 16>, 3, 37, C2, Rr, 2); CK(hipDeviceSynchronize());
            CK(hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost));
            maxd = 0; bad = 0;
            for (size_t i = 0; i < (size_t)Rr * O; ++i) { const double d = std::fabs((double)h1[i] - h2[i]); if (!(d <= 1e-3)) ++bad; if (d > maxd) maxd = d; }
            size_t touched = 0; for (size_t i = (size_t)Rr * O; i < h2.size(); ++i) if (h2[i] == h2[i]) ++touched;   // 0xff.. = NaN pattern
            printf("  nnq<3,spread> G=37 half-first-tile R=%d: max |diff| %.3e, off: %zu, elements written past R: %zu\n", Rr, maxd, bad, touched);
        }
        if (si >= 2 && only != si) continue;          // (extra shapes: correctness only unless selected)
        std::vector<Variant> vs;
        vs.push_back({"baseline gemm_nn_dma (round 2)", [&] { base(C); }, {}});
        vs.push_back({"nnq NS=4 G=512", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 512, C2, R); }, {}});
        auto candl = [&](auto kern, int ns, int G, float* out, int Rr) {
            const size_t lds = (size_t)ns * kNnqStageFloats * 4;
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(G, 1), dim3(320), lds, 0, segs, sh.nseg, sh.F, Rr, Bq, 12, bias, out, O, O, sh.btT, sh.btB, sh.btN);
        };
        {
            CK(hipMemset(C2, 0xff, (size_t)R * O * 4));
            candl(gemm_nnl_kernel<4>, 4, 512, C2, R); CK(hipDeviceSynchronize());
            std::vector<float> h1((size_t)R * O), h2((size_t)R * O);
            CK(hipMemcpy(h1.data(), C, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost));
            double maxd = 0; size_t bad = 0;
            for (size_t i = 0; i < h1.size(); ++i) { const double d = std::fabs((double)h1[i] - h2[i]); if (!(d <= 1e-3)) ++bad; if (d > maxd) maxd = d; }
            printf("  nnl<4> (loader wave) vs baseline, full size: max |diff| %.3e, elements off by > 1e-3: %zu\n", maxd, bad);
        }
        auto candr = [&](int G, float* out, int Rr, int minw = 2, int skew = 0) {
            const size_t lds = ((size_t)4 * 128 * 16 + 192) * 4;
            if (minw == 3) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nnr_kernel<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((gemm_nnr_kernel<4, 3>), dim3(G, 1), dim3(256), lds, 0, segs, sh.nseg, sh.F, Rr, Bq, 12, bias, out, O, O, sh.btT, sh.btB, sh.btN, skew);
            } else {
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nnr_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((gemm_nnr_kernel<4, 2>), dim3(G, 1), dim3(256), lds, 0, segs, sh.nseg, sh.F, Rr, Bq, 12, bias, out, O, O, sh.btT, sh.btB, sh.btN, skew);
            }
        };
        for (int rep = 0; rep < 3; ++rep) {
            const int dbg = 0;
            CK(hipMemset(C2, 0xff, (size_t)R * O * 4));
            candr(512, C2, R, 2, rep * 37); CK(hipDeviceSynchronize());
            std::vector<float> h1((size_t)R * O), h2((size_t)R * O);
            CK(hipMemcpy(h1.data(), C, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost));
            double maxd = 0; size_t bad = 0, badrow_first = 0, nearbias = 0;
            for (size_t i = 0; i < h1.size(); ++i) { const double d = std::fabs((double)h1[i] - h2[i]); if (!(d <= 1e-3)) { if (!bad) badrow_first = i / O; ++bad; } if (d > maxd) maxd = d; }
            printf("  nnr dbg=%d (1: vmcnt(0) behind the tile stores, 2: vmcnt(0) at every chunk) vs baseline: max |diff| %.3e, off: %zu (first bad row %zu)\n", dbg, maxd, bad, badrow_first);
        }
        vs.push_back({"nnr G=512 (weights via registers, 2 WG/CU)", [&] { candr(512, C2, R); }, {}});
        vs.push_back({"nnr G=512 skew 20", [&] { candr(512, C2, R, 2, 20); }, {}});
        vs.push_back({"nnr G=512 skew 40", [&] { candr(512, C2, R, 2, 40); }, {}});
        vs.push_back({"nnr G=512 skew 60", [&] { candr(512, C2, R, 2, 60); }, {}});
        vs.push_back({"nnr G=512 skew 80", [&] { candr(512, C2, R, 2, 80); }, {}});
        vs.push_back({"nnr G=512 skew -40", [&] { candr(512, C2, R, 2, -40); }, {}});
        vs.push_back({"nnr<168 regs, spills> G=768 (3 WG/CU)", [&] { candr(768, C2, R, 3); }, {}});
        vs.push_back({"nnl NS=4 G=512 (loader wave)", [&] { candl(gemm_nnl_kernel<4>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnl NS=3 G=512 (loader wave)", [&] { candl(gemm_nnl_kernel<3>, 3, 512, C2, R); }, {}});
        vs.push_back({"nnl NS=3 G=768 (loader wave, 3 workgroups per CU)", [&] { candl(gemm_nnl_kernel<3>, 3, 768, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 prio alternates per chunk", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 512, C2, R, 8); }, {}});
        vs.push_back({"nnq NS=4 G=512 prio alternates per tile", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 512, C2, R, 16); }, {}});
        vs.push_back({"nnq NS=4 G=512 prio static: younger half high", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 512, C2, R, 24); }, {}});
        vs.push_back({"nnq NS=4 G=512 interleaved + prio per chunk", [&] { cand(gemm_nnq_kernel<4, 64>, 4, 512, C2, R, 8); }, {}});
        vs.push_back({"nnq NS=4 G=512 ABL64 (stores interleaved with the last chunk)", [&] { cand(gemm_nnq_kernel<4, 64>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=3 G=512", [&] { cand(gemm_nnq_kernel<3, 0>, 3, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 first tile 4 + id % 5", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 512, C2, R, 4); }, {}});
        vs.push_back({"nnq NS=4 G=512 first tile 1 + id % 8", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 512, C2, R, 5); }, {}});
        vs.push_back({"nnq NS=4 G=512 first tile 4 + (id >> 3) % 5", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 512, C2, R, 6); }, {}});
        vs.push_back({"nnq NS=3 G=512 first tile 4 + id % 5", [&] { cand(gemm_nnq_kernel<3, 0>, 3, 512, C2, R, 4); }, {}});
        vs.push_back({"nnq NS=4 G=512 half first tile: odd ids", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 512, C2, R, 2); }, {}});
        vs.push_back({"nnq NS=4 G=512 half first tile: id bit 3", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 512, C2, R, 3); }, {}});
        vs.push_back({"nnq NS=3 G=512 half first tile: upper half", [&] { cand(gemm_nnq_kernel<3, 0>, 3, 512, C2, R, 1); }, {}});
        vs.push_back({"nnq NS=4 G=512 spread DMA issue", [&] { cand(gemm_nnq_kernel<4, 16>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 spread + upper half", [&] { cand(gemm_nnq_kernel<4, 16>, 4, 512, C2, R, 1); }, {}});
        vs.push_back({"nnq NS=4 G=512 ABL4 (A cache-hot)", [&] { cand(gemm_nnq_kernel<4, 4>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 ABL8 (B cache-hot)", [&] { cand(gemm_nnq_kernel<4, 8>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 ABL12 (A+B cache-hot)", [&] { cand(gemm_nnq_kernel<4, 12>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 ABL13 (hot, no stores)", [&] { cand(gemm_nnq_kernel<4, 13>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 ABL32 (C cache-resident)", [&] { cand(gemm_nnq_kernel<4, 32>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 ABL1 (no C stores)", [&] { cand(gemm_nnq_kernel<4, 1>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 ABL3 (no DMA, no stores)", [&] { cand(gemm_nnq_kernel<4, 3>, 4, 512, C2, R); }, {}});
        run_variants(vs, rounds, fl);
        {   // cycle probe (wave 0 of every workgroup)
            long long* pr; CK(hipMalloc(&pr, 512 * 10 * 8)); CK(hipMemset(pr, 0, 512 * 10 * 8));
            for (int abl = 0; abl < 4; ++abl) {
                const size_t lds = (size_t)4 * kNnqStageFloats * 4;
                if (abl == 0 || abl == 3) { hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nnq_kernel<4, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipLaunchKernelGGL((gemm_nnq_kernel<4, 128>), dim3(512, 1), dim3(256), lds, 0, segs, sh.nseg, sh.F, R, Bq, 12, bias, C2, O, O, sh.btT, sh.btB, sh.btN, abl == 3 ? 8 : 0, pr); }
                else if (abl == 2) { hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nnq_kernel<4, 192>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipLaunchKernelGGL((gemm_nnq_kernel<4, 192>), dim3(512, 1), dim3(256), lds, 0, segs, sh.nseg, sh.F, R, Bq, 12, bias, C2, O, O, sh.btT, sh.btB, sh.btN, 0, pr); }
                else { hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nnq_kernel<4, 129>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipLaunchKernelGGL((gemm_nnq_kernel<4, 129>), dim3(512, 1), dim3(256), lds, 0, segs, sh.nseg, sh.F, R, Bq, 12, bias, C2, O, O, sh.btT, sh.btB, sh.btN, 0, pr); }
                CK(hipDeviceSynchronize());
                std::vector<long long> h(512 * 10); CK(hipMemcpy(h.data(), pr, h.size() * 8, hipMemcpyDeviceToHost));
                double a[10] = {0}; for (int b = 0; b < 512; ++b) for (int i = 0; i < 10; ++i) a[i] += (double)h[b * 10 + i] / 512;
                printf("  probe %s: per workgroup: barrier wait after an epilogue %.0f / %.0f / %.0f cycles (iterations +1/+2/+3), other waits %.0f avg over %.0f; iteration (wait + MFMAs) avg %.0f cycles; epilogue %.0f cycles avg over %.0f\n",
                       abl == 1 ? "(no C stores)" : abl == 2 ? "(interleaved stores)" : abl == 3 ? "(burst stores, prio per chunk)" : "(burst stores)", a[0] / a[7], a[1] / a[7], a[2] / a[7], a[3] / a[4], a[4], a[5] / (a[4] + 3 * a[7]), a[6] / a[7], a[7]);
                {   long long t0 = h[8], t1 = h[9]; for (int b = 0; b < 512; ++b) { t0 = std::min(t0, h[b * 10 + 8]); t1 = std::max(t1, h[b * 10 + 9]); }
                    std::vector<double> dur, st, en; for (int b = 0; b < 512; ++b) { dur.push_back((h[b * 10 + 9] - h[b * 10 + 8]) / 100.0); st.push_back((h[b * 10 + 8] - t0) / 100.0); en.push_back((h[b * 10 + 9] - t0) / 100.0); }
                    std::sort(dur.begin(), dur.end()); std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
                    printf("      workgroup durations us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f | starts: median %.1f max %.1f | ends: p10 %.1f median %.1f max %.1f (span %.1f us)\n",
                           dur[0], dur[51], dur[256], dur[460], dur[511], st[256], st[511], en[51], en[256], en[511], (t1 - t0) / 100.0);
                    // by XCD (block id % 8) and by half
                    for (int x = 0; x < 8; ++x) { double m = 0; for (int b = x; b < 512; b += 8) m = std::max(m, (h[b * 10 + 9] - t0) / 100.0); printf("%s xcd%d %.1f", x ? "" : "      last end by id%8:", x, m); }
                    double lo = 0, hi = 0; for (int b = 0; b < 256; ++b) lo = std::max(lo, (h[b * 10 + 9] - t0) / 100.0); for (int b = 256; b < 512; ++b) hi = std::max(hi, (h[b * 10 + 9] - t0) / 100.0);
                    printf(" | ids < 256: %.1f, >= 256: %.1f\n", lo, hi);
                    for (int hf = 0; hf < 2; ++hf) { double g[8] = {0}; for (int b = hf * 256; b < hf * 256 + 256; ++b) for (int i = 0; i < 8; ++i) g[i] += (double)h[b * 10 + i] / 256;
                        printf("      ids %s: waits after epilogue %.0f / %.0f / %.0f, other waits %.0f, iteration avg %.0f, epilogue %.0f cycles\n", hf ? ">= 256" : "<  256", g[0] / g[7], g[1] / g[7], g[2] / g[7], g[3] / g[4], g[5] / (g[4] + 3 * g[7]), g[6] / g[7]); } }
            }
        }
    }

    // ================================= TN (weight gradients) ==================================================
    if (only < 0 || only >= 10) {
        float *dYb, *P1, *P2;
        CK(hipMalloc(&dYb, (size_t)R * 192 * 4)); fill_rand(dYb, (size_t)R * 192, 9);
        const size_t pmax = (size_t)1024 * 320 * 192;
        CK(hipMalloc(&P1, pmax * 4)); CK(hipMalloc(&P2, pmax * 4));
        struct TShape { const char* name; int nseg, F, ycol0, Ov, btT, btB, btN; };
        const TShape ts[] = {{"x-part layer 1 (K=192, O=192)", 3, 64, 0, 192, 0, 0, 0}, {"h-gate (K=192, O=128)", 3, 64, 0, 128, 0, 0, 0},
                             {"h-cand (K=192, O=64)", 3, 64, 128, 64, 0, 0, 0}, {"x-part layer 0 (K=300, O=192, batch-major rows)", 3, 100, 0, 192, 60, 256, 19},
                             {"5 planes x-part / dX (K=320, O=192)", 5, 64, 0, 192, 0, 0, 0}, {"5 planes h-gate (K=320, O=128)", 5, 64, 0, 128, 0, 0, 0},
                             {"5 planes h-cand (K=320, O=64)", 5, 64, 128, 64, 0, 0, 0}};
        for (int ti = 0; ti < 7; ++ti) {
            if (ti >= 4 && only != 10 + ti) continue;
            if (only >= 10 && only != 10 + ti) continue;
            const TShape t = ts[ti];
            const int K = t.nseg * t.F;
            SegPtrs segs{}; for (int m = 0; m < t.nseg; ++m) segs.p[m] = A + (size_t)m * R * t.F;
            const double fl = 2.0 * R * (double)K * t.Ov;
            printf("== TN %s: R=%d\n", t.name, R);
            // round-2 kernel with its own split plan (tn_split of api.cpp: ~768 workgroups)
            const int kblocks = (K + 63) / 64;
            int nsplit_o = (768 + kblocks - 1) / kblocks, rps_o = ((R + nsplit_o - 1) / nsplit_o + 31) / 32 * 32;
            nsplit_o = (R + rps_o - 1) / rps_o; if (nsplit_o >= 8) nsplit_o = (nsplit_o + 7) / 8 * 8;
            auto old_tn = [&](float* out) {
                auto go = [&](auto kern, int rc, int otile) {
                    const size_t lds = 2 * (size_t)(rc * 64 + rc * otile) * 4;
                    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipLaunchKernelGGL(kern, dim3(kblocks, nsplit_o), dim3(256), lds, 0, segs, t.nseg, t.F, R, dYb, 192, t.ycol0, t.Ov, out, rps_o, t.btT, t.btB, t.btN, 1);
                };
                if (t.Ov > 128) go(gemm_tn_dma_kernel<2, 6, 16, 2>, 16, 192);
                else if (t.Ov > 64) go(gemm_tn_dma_kernel<2, 4, 32, 2>, 32, 128);
                else go(gemm_tn_dma_kernel<2, 2, 32, 2>, 32, 64);
            };
            auto new_tn = [&](auto kern, int kt, int ot, int rc, int G, float* out, int Rr, int flags = 0) {
                const int nkb = (K + 32 * kt - 1) / (32 * kt);
                int nsplit = G / nkb; if (nsplit < 1) nsplit = 1;
                const int rps = ((Rr + nsplit - 1) / nsplit + rc - 1) / rc * rc;
                nsplit = (Rr + rps - 1) / rps;
                const size_t lds = 3 * (size_t)(rc * 32 * (kt + ot)) * 4;
                hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(kern, dim3(nkb, nsplit), dim3(256), lds, 0, segs, t.nseg, t.F, Rr, dYb, 192, t.ycol0, t.Ov, out, rps, t.btT, t.btB, t.btN, flags);
                return nsplit;
            };
            auto launch_new_t = [&](auto TLC, int G, float* out, int Rr, int variant) {
                constexpr bool TL = decltype(TLC)::value != 0;
                const int fl_ = variant >= 10 ? variant - 9 : 0; if (variant >= 10) variant = 0;
                if (ti == 0) return variant == 0 ? new_tn(gemm_tnq_kernel<6, 6, 16, false, true, TL>, 6, 6, 16, G, out, Rr, fl_) : new_tn(gemm_tnq_kernel<4, 6, 16, false, true, TL>, 4, 6, 16, G, out, Rr, fl_);
                if (ti == 1) return variant == 0 ? new_tn(gemm_tnq_kernel<6, 4, 16, false, true, TL>, 6, 4, 16, G, out, Rr, fl_) : new_tn(gemm_tnq_kernel<6, 4, 32, false, true, TL>, 6, 4, 32, G, out, Rr, fl_);
                if (ti == 2) return variant == 0 ? new_tn(gemm_tnq_kernel<6, 2, 16, false, true, TL>, 6, 2, 16, G, out, Rr, fl_) : new_tn(gemm_tnq_kernel<6, 2, 32, false, true, TL>, 6, 2, 32, G, out, Rr, fl_);
                // five planes of 64: generic blocks of 160 (exact) against whole-plane blocks of 192 (one padding plane)
                if (ti == 4) return variant == 0 ? new_tn(gemm_tnq_kernel<5, 6, 16, false, false, TL>, 5, 6, 16, G, out, Rr, fl_) : new_tn(gemm_tnq_kernel<6, 6, 16, false, true, TL>, 6, 6, 16, G, out, Rr, fl_);
                if (ti == 5) return variant == 0 ? new_tn(gemm_tnq_kernel<5, 4, 16, false, false, TL>, 5, 4, 16, G, out, Rr, fl_) : new_tn(gemm_tnq_kernel<6, 4, 16, false, true, TL>, 6, 4, 16, G, out, Rr, fl_);
                if (ti == 6) return variant == 0 ? new_tn(gemm_tnq_kernel<5, 2, 16, false, false, TL>, 5, 2, 16, G, out, Rr, fl_) : new_tn(gemm_tnq_kernel<6, 2, 16, false, true, TL>, 6, 2, 16, G, out, Rr, fl_);
                return variant == 0 ? new_tn(gemm_tnq_kernel<5, 6, 16, true, false, TL>, 5, 6, 16, G, out, Rr, fl_) : new_tn(gemm_tnq_kernel<4, 6, 16, true, false, TL>, 4, 6, 16, G, out, Rr, fl_);
            };
            auto launch_new = [&](int G, float* out, int Rr, int variant) { return Rr % 32 == 0 ? launch_new_t(IntC<0>(), G, out, Rr, variant) : launch_new_t(IntC<1>(), G, out, Rr, variant); };
            // correctness: sum of the partials over the splits, both kernels, every element; also a ragged R
            for (int pass = 0; pass < 2; ++pass) {
                const int Rr = pass == 0 ? R : (t.btT > 0 ? R : 100003);
                if (pass == 1 && t.btT > 0) continue;
                std::vector<double> ref((size_t)K * t.Ov, 0.0), got((size_t)K * t.Ov, 0.0);
                if (pass == 0) {
                    CK(hipMemset(P1, 0, pmax * 4)); old_tn(P1); CK(hipDeviceSynchronize());
                    std::vector<float> h((size_t)nsplit_o * K * t.Ov); CK(hipMemcpy(h.data(), P1, h.size() * 4, hipMemcpyDeviceToHost));
                    for (int sp = 0; sp < nsplit_o; ++sp) for (size_t e = 0; e < ref.size(); ++e) ref[e] += h[(size_t)sp * ref.size() + e];
                } else {                                            // host reference on the first Rr rows (a few columns of K only would hide bugs: do all, fp64)
                    std::vector<float> ha((size_t)Rr * t.F * t.nseg), hy((size_t)Rr * 192);
                    for (int m = 0; m < t.nseg; ++m) CK(hipMemcpy(ha.data() + (size_t)m * Rr * t.F, A + (size_t)m * R * t.F, (size_t)Rr * t.F * 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hy.data(), dYb, hy.size() * 4, hipMemcpyDeviceToHost));
                    for (int r = Rr - 3000; r < Rr; ++r)            // only the last rows differ from pass 0's coverage: check the tail handling
                        for (int k = 0; k < K; ++k) { const double av = ha[(size_t)(k / t.F) * Rr * t.F + (size_t)r * t.F + k % t.F];
                            for (int o = 0; o < t.Ov; ++o) ref[(size_t)k * t.Ov + o] += av * hy[(size_t)r * 192 + t.ycol0 + o]; }
                }
                CK(hipMemset(P2, 0xff, pmax * 4));
                int ns;
                if (pass == 0) ns = launch_new(512, P2, Rr, 0);
                else {          // rows [Rr-3000, Rr) only: emulate with a one-split launch over a shifted view is not possible -> compare full sums instead
                    ns = launch_new(37, P2, Rr, 0);
                }
                CK(hipDeviceSynchronize());
                std::vector<float> h((size_t)ns * K * t.Ov); CK(hipMemcpy(h.data(), P2, h.size() * 4, hipMemcpyDeviceToHost));
                if (pass == 1) {       // subtract the contribution of rows [0, Rr-3000) computed by the same kernel on an R that is a multiple of 16
                    std::fill(got.begin(), got.end(), 0.0);
                    for (int sp = 0; sp < ns; ++sp) for (size_t e = 0; e < got.size(); ++e) got[e] += h[(size_t)sp * got.size() + e];
                    const int Rh = Rr - 3000 - ((Rr - 3000) % 16 ? 0 : 0);
                    CK(hipMemset(P2, 0xff, pmax * 4));
                    // head rows through the old kernel is not available for arbitrary R; use the new kernel with a different split count (G = 5)
                    const int ns2 = launch_new(5, P2, Rh, 0); CK(hipDeviceSynchronize());
                    std::vector<float> h2((size_t)ns2 * K * t.Ov); CK(hipMemcpy(h2.data(), P2, h2.size() * 4, hipMemcpyDeviceToHost));
                    for (int sp = 0; sp < ns2; ++sp) for (size_t e = 0; e < got.size(); ++e) got[e] -= h2[(size_t)sp * got.size() + e];
                } else {
                    for (int sp = 0; sp < ns; ++sp) for (size_t e = 0; e < got.size(); ++e) got[e] += h[(size_t)sp * got.size() + e];
                }
                double maxd = 0, maxv = 0; size_t nan = 0;
                for (size_t e = 0; e < ref.size(); ++e) { const double d = std::fabs(ref[e] - got[e]); if (!(d == d)) ++nan; if (d > maxd) maxd = d; if (std::fabs(ref[e]) > maxv) maxv = std::fabs(ref[e]); }
                printf("  tnq vs %s (R=%d, %d splits): max |diff| %.3e (max |value| %.1f), NaN %zu\n", pass == 0 ? "round-2 kernel" : "fp64 host sum of the last 3000 rows", Rr, ns, maxd, maxv, nan);
            }
            std::vector<Variant> vs;
            vs.push_back({"baseline gemm_tn_dma (round 2)", [&] { old_tn(P1); }, {}});
            vs.push_back({"tnq G=512", [&] { launch_new(512, P2, R, 0); }, {}});
            vs.push_back({"tnq G=1024", [&] { launch_new(1024, P2, R, 0); }, {}});
            vs.push_back({"tnq G=512 no partial stores", [&] { launch_new(512, P2, R, 10); }, {}});
            vs.push_back({"tnq G=512 no DMA after the prologue", [&] { launch_new(512, P2, R, 13); }, {}});
            vs.push_back({"tnq G=512 no DMA, no partial stores", [&] { launch_new(512, P2, R, 14); }, {}});
            vs.push_back({"tnq G=512 partials into 8 cache-resident slots", [&] { launch_new(512, P2, R, 11); }, {}});
            vs.push_back({"tnq G=448", [&] { launch_new(448, P2, R, 0); }, {}});
            vs.push_back({"tnq G=256", [&] { launch_new(256, P2, R, 0); }, {}});
            vs.push_back({ti >= 4 ? "tnq whole planes KT=6 G=512" : ti == 0 ? "tnq KT=4 (2 k-blocks, 2nd half empty) G=512" : ti == 3 ? "tnq KT=4 (3 k-blocks) G=512" : "tnq RC=32 G=512", [&] { launch_new(512, P2, R, 1); }, {}});
            vs.push_back({ti >= 4 ? "tnq whole planes KT=6 G=1024" : ti == 0 ? "tnq KT=4 (2 k-blocks, 2nd half empty) G=1024" : ti == 3 ? "tnq KT=4 (3 k-blocks) G=1024" : "tnq RC=32 G=1024", [&] { launch_new(1024, P2, R, 1); }, {}});
            run_variants(vs, rounds, fl);
        }
    }
    return 0;
}
