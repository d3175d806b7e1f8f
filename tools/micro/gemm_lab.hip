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
#include "../../eeg_gnn_ssl_amd/csrc/kernels_gemm_q.h"
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
    CK(hipMalloc(&A, (size_t)3 * R * 100 * 4)); CK(hipMalloc(&Bold, 320 * 192 * 4)); CK(hipMalloc(&Bq, 320 * 192 * 4));
    CK(hipMalloc(&C, (size_t)R * O * 4)); CK(hipMalloc(&C2, (size_t)R * O * 4)); CK(hipMalloc(&bias, 192 * 4));
    fill_rand(A, (size_t)3 * R * 100, 1); fill_rand(bias, 192, 3);

    struct Shape { const char* name; int nseg, F; int btT, btB, btN; };
    const Shape shapes[] = {{"layer-1 x-part / dX (K=192)", 3, 64, 0, 0, 0}, {"layer-0 x-part (K=300, batch-major rows)", 3, 100, 60, 256, 19}};
    for (int si = 0; si < 2; ++si) {
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
        auto cand = [&](auto kern, int ns, int G, float* out, int Rr) {
            const size_t lds = (size_t)ns * kNnqStageFloats * 4;
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(G, 1), dim3(256), lds, 0, segs, sh.nseg, sh.F, Rr, Bq, 12, bias, out, O, O, sh.btT, sh.btB, sh.btN);
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
            cand(gemm_nnq_kernel<3, 0>, 3, 37, C2, Rr); CK(hipDeviceSynchronize());
            CK(hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost));
            maxd = 0; bad = 0;
            for (size_t i = 0; i < (size_t)Rr * O; ++i) { const double d = std::fabs((double)h1[i] - h2[i]); if (!(d <= 1e-3)) ++bad; if (d > maxd) maxd = d; }
            size_t touched = 0; for (size_t i = (size_t)Rr * O; i < h2.size(); ++i) if (h2[i] == h2[i]) ++touched;   // 0xff.. = NaN pattern
            printf("  nnq<3> G=37 R=%d: max |diff| %.3e, off: %zu, elements written past R: %zu\n", Rr, maxd, bad, touched);
        }
        std::vector<Variant> vs;
        vs.push_back({"baseline gemm_nn_dma (round 2)", [&] { base(C); }, {}});
        vs.push_back({"nnq NS=4 G=512", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=3 G=512", [&] { cand(gemm_nnq_kernel<3, 0>, 3, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=256", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 256, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=768 (3rd WG per CU queues)", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 768, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=2280 (one tile per WG)", [&] { cand(gemm_nnq_kernel<4, 0>, 4, 2280, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 ABL1 (no C stores)", [&] { cand(gemm_nnq_kernel<4, 1>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 ABL2 (no DMA in loop)", [&] { cand(gemm_nnq_kernel<4, 2>, 4, 512, C2, R); }, {}});
        vs.push_back({"nnq NS=4 G=512 ABL3 (no DMA, no stores)", [&] { cand(gemm_nnq_kernel<4, 3>, 4, 512, C2, R); }, {}});
        run_variants(vs, rounds, fl);
    }
    return 0;
}
