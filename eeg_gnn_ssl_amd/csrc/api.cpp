// C ABI of libeeg_dcrnn_hip.so (see include/eeg_dcrnn.h): argument checking + kernel orchestration.
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <string>
#include <vector>

#include "../../include/eeg_dcrnn.h"
#include "../../include/eeg_dcrnn_dev.h"
#include "../../include/eeg_dcrnn_prof.h"
#include "kernels_decoder.h"
#include "kernels_diffuse.h"
#include "kernels_feat.h"
#include "kernels_gemm.h"
#include "kernels_gemm_bf.h"
#include "kernels_graph.h"
#include "kernels_head.h"
#include "kernels_pack.h"
#include "spec_common.h"
#include "spec_launch.h"
#include "kernels_tail.h"
#include "prof.h"
#include "seq_launch.h"
#include "gemmq_launch.h"

#include <string>
#include <vector>


namespace {

thread_local char g_err[512] = "";
// Development aids exist only in the dev build (make dev / the test emulator: -DEEG_DEV, declared in
// include/eeg_dcrnn_dev.h).  In the product build the knobs are compile-time zeros and there is no probe.
#if defined(EEG_DEV)
long long* g_seq_probe = nullptr;  // see eeg_dcrnn_set_seq_probe
int g_tune[24] = {0};               // see eeg_dcrnn_set_tuning
#else
constexpr int g_tune[24] = {0};
#endif

std::atomic<long long*> g_clock_samples{nullptr};   // eeg_dcrnn_prof_clock_samples (measurement hook, like the event recorder)
// the `probe` argument of the recurrent launches: the dev build's phase probe (which selects the probe instantiations), else the
// product's clock-sample buffer (4 x int64; dev builds leave it alone: a non-null probe means 32 slots per wave there).
// A launch that is being CAPTURED into a graph never gets the clock-sample buffer: the graph would keep the pointer and go on
// adding to it on every replay, after the buffer was disarmed and freed.
long long* seq_probe_arg(hipStream_t st) {
#if defined(EEG_DEV)
    (void)st;
    return g_seq_probe;
#else
    long long* p = g_clock_samples.load(std::memory_order_acquire);
    if (p != nullptr && eeg::platform_stream_is_capturing(st)) return nullptr;
    return p;
#endif
}

// the dev build's phase probe is armed (its instantiations exist for the register-resident kernels only)
bool phase_probe_armed() {
#if defined(EEG_DEV)
    return g_seq_probe != nullptr;
#else
    return false;
#endif
}

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}
inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

using namespace eeg;

bool h_supported(int H) { return H == 16 || H == 32 || H == 64; }
bool m_supported(int M) { return M == 1 || M == 2 || M == 3 || M == 4 || M == 5 || M == 7; }

int check_dims(int N, int H, int Fin, int M) {
    if (N < 1 || N > kMaxNodes) return fail("num_nodes=%d unsupported (1..%d)", N, kMaxNodes);
    if (!h_supported(H)) return fail("rnn_units=%d unsupported (16, 32 or 64)", H);
    if (Fin < 4 || Fin % 4 != 0) return fail("per-node input dim=%d unsupported (must be a positive multiple of 4)", Fin);
    if (!m_supported(M)) return fail("num hop matrices M=%d unsupported (1,2,3,4,5,7)", M);
    // LDS of the BPTT kernel (SeqGeom::bwd_lds_floats): the widest case (H=64, M=7) only fits montages of <= 20 nodes
    const int ka = M * H, rows = N <= 20 ? 20 : 32;
    const size_t bwd = ((size_t)(M - 1) * kPFloats + (size_t)rows * (lds_stride_x(ka) + lds_stride_x(2 * ka))) * sizeof(float);
    if (bwd > kMaxLdsBytes)
        return fail("rnn_units=%d with %d hop matrices and %d nodes needs %zu KB of LDS for the backward pass (160 available)",
                    H, M, N, bwd / 1024);
    return 0;
}

// CUs of the current device (the persistent GEMMs size their grids by it); queried once per process
int num_cus() { return platform_num_cus(); }

// fewest rows the persistent round-3 GEMMs are used for (below it the ramp of their 2-per-CU grid costs more than the
// round-2 kernels' many small workgroups); dev knob 2, bits 2..: rows per CU instead of the default
int q_min_rows() { return ((g_tune[2] >> 2) > 0 ? (g_tune[2] >> 2) : 256) * num_cus(); }

// ---- GEMM dispatch ---------------------------------------------------------------------------
template <int NCTW, int KC>
int run_nn(const SegPtrs& segs, int nseg, int F, int R, const float* Bp, int nct_total, const float* bias,
           float* C, int ldc, int O, hipStream_t st, const char* tag) {
    constexpr int KCS = lds_stride(KC), NB = 2 * NCTW;
    const size_t lds = 2 * (size_t)(128 * KCS + (KC / 4) * NB * 64) * sizeof(float);
    EEG_SET_MAX_LDS((gemm_nn_kernel<NCTW, KC>), lds);
    dim3 grid(ceil_div(R, 128), ceil_div(nct_total, NB));
    EEG_LAUNCH_P(tag, (gemm_nn_kernel<NCTW, KC>), grid, dim3(256), lds, st, segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O);
    return check_launch("gemm_nn");
}
// the A segments of a GEMM are batch-major (B clips x T steps x N nodes): the DMA kernels read them through a row map
struct BtMap { int T = 0, B = 0, N = 0; };
template <int NCTW, int KC, int MINB = 2>
int run_nn_dma(const SegPtrs& segs, int nseg, int F, int R, const float* Bp, int nct_total, const float* bias,
               float* C, int ldc, int O, hipStream_t st, const char* tag, BtMap bt) {
    constexpr int NB = 2 * NCTW;
    const size_t lds = 2 * (size_t)(128 * KC + (KC / 4) * NB * 64) * sizeof(float);
    EEG_SET_MAX_LDS((gemm_nn_dma_kernel<NCTW, KC, MINB>), lds);
    dim3 grid(ceil_div(R, 128), ceil_div(nct_total, NB));
    EEG_LAUNCH_P(tag, (gemm_nn_dma_kernel<NCTW, KC, MINB>), grid, dim3(256), lds, st, segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O, bt.T, bt.B, bt.N);
    return check_launch("gemm_nn_dma");
}
bool nn_dma_ok(int F, int R, int ldc) { return g_tune[0] == 0 && ldc % 4 == 0 && (double)R * F < 4.0e9 && (F % 16 == 0 || F % 20 == 0); }
template <int NCTW>
int run_nn_kc(const SegPtrs& segs, int nseg, int F, int R, const float* Bp, int nct_total, const float* bias,
              float* C, int ldc, int O, hipStream_t st, const char* tag, BtMap bt) {
    if (nn_dma_ok(F, R, ldc)) {                                      // LDS-DMA staging (default)
        if (F % 16 == 0) return run_nn_dma<NCTW, 16>(segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O, st, tag, bt);
        if (F % 20 == 0) return run_nn_dma<NCTW, 20>(segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O, st, tag, bt);
    }
    if (bt.T > 0) return fail("gemm_nn: a batch-major segment needs the LDS-DMA kernel (F=%d)", F);
    if (F % 32 == 0) return run_nn<NCTW, 32>(segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O, st, tag);
    if (F % 20 == 0) return run_nn<NCTW, 20>(segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O, st, tag);
    if (F % 16 == 0) return run_nn<NCTW, 16>(segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O, st, tag);
    return run_nn<NCTW, 4>(segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O, st, tag);
}
// C[R x O] = [segments] @ packed B (nct_total col tiles) + bias
// Bq: the same right-hand side in the quad order of gemm_nnr_kernel (kernels_pack.h: bxq / bxtq), or NULL.  The round-3
// kernel takes the launch when it covers the shape and every one of its 2-per-CU workgroups gets at least two 128-row
// tiles (dev knob 2 bit 0 = 1: never)
int gemm_nn(const SegPtrs& segs, int nseg, int F, int R, const float* Bp, int nct_total, const float* bias,
            float* C, int ldc, int O, hipStream_t st, const char* tag = "gemm_nn", BtMap bt = BtMap(), const float* Bq = nullptr) {
    if (Bq != nullptr && (g_tune[2] & 1) == 0 && g_tune[0] == 0 && R >= q_min_rows() && nnq_supported(nseg, F, R, nct_total, ldc, O)) {
        if (launch_nnq(segs, nseg, F, R, Bq, nct_total, bias, C, ldc, O, bt.T, bt.B, bt.N, num_cus(), st, tag)) return fail("gemm_nnq: launch failed");
        return check_launch("gemm_nnq");
    }
    // few row blocks (per-step decoder GEMMs): narrower column blocks fill more CUs
    if (nct_total <= 4 || ceil_div(R, 128) * ceil_div(nct_total, 12) < 160)
        return run_nn_kc<2>(segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O, st, tag, bt);
    // column block of 12, 10 or 8 tiles, whichever leaves the fewest padding tiles (20 tiles = dX at M = 5: 2 x 10
    // instead of 2 x 12 with a sixth of the MFMAs on padding)
    const int pad6 = round_up(nct_total, 12) - nct_total, pad5 = round_up(nct_total, 10) - nct_total, pad4 = round_up(nct_total, 8) - nct_total;
    if (pad5 < pad6 && pad5 <= pad4 && F % 16 == 0 && nn_dma_ok(F, R, ldc))
        return run_nn_dma<5, 16>(segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O, st, tag, bt);
    if (pad4 < pad6 && nn_dma_ok(F, R, ldc)) return run_nn_kc<4>(segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O, st, tag, bt);
    return run_nn_kc<6>(segs, nseg, F, R, Bp, nct_total, bias, C, ldc, O, st, tag, bt);
}

// opt-in three-term bf16 split of an NN GEMM (kernels_gemm_bf.h).  Returns -1 when the shape has no instantiation (the caller
// then takes the fp32 kernel), 0 ok, 1 error.
template <int NTB>
int run_nn_bf3(const SegPtrs& segs, int nseg, int F, int R, const unsigned short* Wp, int nct_total, const float* bias, float* C,
               int ldc, int O, hipStream_t st, const char* tag, BtMap bt) {
    const size_t lds = 2 * 3 * (size_t)NTB * 1024;
    EEG_SET_MAX_LDS((gemm_nn_bf3_kernel<NTB>), lds);
    EEG_LAUNCH_P(tag, (gemm_nn_bf3_kernel<NTB>), dim3(ceil_div(R, 128), nct_total / NTB), dim3(256), lds, st, segs, nseg, F, R, Wp, nct_total,
                 bias, C, ldc, O, bt.T, bt.B, bt.N);
    return check_launch("gemm_nn_bf3");
}
int gemm_nn_bf3(const SegPtrs& segs, int nseg, int F, int R, const unsigned short* Wp, int nct_total, const float* bias, float* C,
                int ldc, int O, hipStream_t st, const char* tag, BtMap bt) {
    if (F % 4 != 0 || ldc % 4 != 0 || O % 4 != 0 || (double)R * F >= 4.0e9) return -1;
    if (nct_total % 12 == 0) return run_nn_bf3<12>(segs, nseg, F, R, Wp, nct_total, bias, C, ldc, O, st, tag, bt);
    if (nct_total % 10 == 0) return run_nn_bf3<10>(segs, nseg, F, R, Wp, nct_total, bias, C, ldc, O, st, tag, bt);
    if (nct_total % 8 == 0) return run_nn_bf3<8>(segs, nseg, F, R, Wp, nct_total, bias, C, ldc, O, st, tag, bt);
    return -1;
}

template <int NCTW>
int run_tn(const SegPtrs& segs, int nseg, int F, int R, const float* dY, int ldy, int ycol0, int O,
           float* partial, int nsplit, int rows_per_split, hipStream_t st, const char* tag) {
    constexpr int OT = 2 * NCTW * 16;
    constexpr int YS = OT + ((16 - (OT % 32)) + 32) % 32;
    const size_t lds = 2 * (size_t)(32 * 80 + 32 * YS) * sizeof(float);
    EEG_SET_MAX_LDS((gemm_tn_kernel<NCTW>), lds);
    dim3 grid(nseg * ceil_div(F, 64), nsplit);
    // same XCD-aware placement as the DMA kernel (run_tn_dma); in this register-staged kernel it measured slower in round 1
    // (1.19 vs 1.04 ms per step), so here it stays a dev knob (4 = 2)
    const int remap = (g_tune[4] == 2 && nsplit % 8 == 0 && grid.x > 1) ? 1 : 0;
    EEG_LAUNCH_P(tag, (gemm_tn_kernel<NCTW>), grid, dim3(256), lds, st, segs, nseg, F, R, dY, ldy, ycol0, O, partial, rows_per_split, remap);
    return check_launch("gemm_tn");
}
// k-block width of the DMA TN kernel.  128-wide blocks halve the re-reads of dY (PMC: 862 -> ~600 MB per
// launch) but measured SLOWER (gemm_tn 1.08-1.15 vs 0.92 ms/step, cfg2): the re-reads are served by the
// Infinity Cache, and the wider tile costs occupancy.
constexpr int kTnKbw = 64;
template <int KTW, int NCTW, int RC, int WK = 2>
int run_tn_dma(const SegPtrs& segs, int nseg, int F, int R, const float* dY, int ldy, int ycol0, int O,
               float* partial, int nsplit, int rows_per_split, hipStream_t st, const char* tag, BtMap bt) {
    constexpr int OT = 2 * NCTW * 16, KBW = 16 * KTW * WK;
    const size_t lds = 2 * (size_t)(RC * KBW + RC * OT) * sizeof(float);
    EEG_SET_MAX_LDS((gemm_tn_dma_kernel<KTW, NCTW, RC, WK>), lds);
    dim3 grid(ceil_div(nseg * F, KBW), nsplit);
    // all k-blocks of a row split on ONE XCD (they read the same dY rows): PMC traffic of the class 831 -> 432 MB per launch
    // (1.93x -> 1.00x algorithmic) at unchanged time (dev knob 4 = 1 switches the placement off)
    const int remap = (g_tune[4] == 0 && nsplit % 8 == 0 && grid.x > 1) ? 1 : 0;
    EEG_LAUNCH_P(tag, (gemm_tn_dma_kernel<KTW, NCTW, RC, WK>), grid, dim3(128 * WK), lds, st, segs, nseg, F, R, dY, ldy, ycol0, O, partial, rows_per_split, bt.T, bt.B, bt.N, remap);
    return check_launch("gemm_tn_dma");
}

// dev knob 14: 8-wave / 128-column k-blocks for dY tiles of this width and up (0 = never); knob 15: their workgroup target
int tn_wide_from() { return g_tune[14]; }
bool tn_dma_ok(int F, int O) { return g_tune[1] == 0 && O > 32 && O % 4 == 0 && F % 4 == 0; }
int tn_split(int nseg, int F, int R, int O, int* rows_per_split) {
    const bool dma = tn_dma_ok(F, O);
    const bool wide = dma && tn_wide_from() > 0 && O >= tn_wide_from();
    const int blocks = dma ? ceil_div(nseg * F, wide ? 2 * kTnKbw : kTnKbw) : nseg * ceil_div(F, 64);
    // ~3 workgroups per CU (wide: 2) per 400 k rows: measured, R = 291 840 rows (cfg2/3/4): 768 workgroups best (1152: +12 %,
    // 1536: +1..10 %); R = 583 680 (cfg5): 1536 best (768: +4 %, and +20 % on the h-gate shape with the XCD placement)
    const int target = (wide ? 512 : 768) * ceil_div(R, 400000);
    int nsplit = ceil_div(g_tune[15] > 0 ? g_tune[15] : target, blocks);
    int rps = round_up(ceil_div(R, nsplit), 32);
    if (rps < 128) rps = 128;
    nsplit = ceil_div(R, rps);
    if (nsplit >= 8) nsplit = round_up(nsplit, 8);   // multiple of 8: XCD-aware k-block placement (trailing splits may be empty)
    *rows_per_split = rps;
    return nsplit;
}
// partial[nsplit][nseg*F][O] = per-split A^T dY[:, ycol0:ycol0+O]
int gemm_tn(const SegPtrs& segs, int nseg, int F, int R, const float* dY, int ldy, int ycol0, int O,
            float* partial, int nsplit, int rows_per_split, hipStream_t st, const char* tag = "gemm_tn", BtMap bt = BtMap(),
            const TnqPlan* q = nullptr) {
    if (q != nullptr && q->ok) {                         // round-3 kernel (nsplit / rows_per_split are the plan's)
        if (launch_tnq(*q, segs, nseg, F, R, dY, ldy, ycol0, O, partial, bt.T, bt.B, bt.N, st, tag)) return fail("gemm_tnq: launch failed");
        return check_launch("gemm_tnq");
    }
    if (tn_dma_ok(F, O) && rows_per_split % 32 == 0 && R >= 1) {   // LDS-DMA staging (default)
#define EEG_TN(NCTW, RC) run_tn_dma<2, NCTW, RC>(segs, nseg, F, R, dY, ldy, ycol0, O, partial, nsplit, rows_per_split, st, tag, bt)
        // row-chunk depth: the 192-column tile stages 16 rows at a time (32 KB of LDS per workgroup -> 4 workgroups
        // per CU instead of 2 with 32-row stages: -3.5 % on that shape); the narrower tiles are better off with 32
        // rows (measured both ways); 8-row stages are 15 % slower
#define EEG_TNW(NCTW, RC) run_tn_dma<2, NCTW, RC, 4>(segs, nseg, F, R, dY, ldy, ycol0, O, partial, nsplit, rows_per_split, st, tag, bt)
        const bool wide = tn_wide_from() > 0 && O >= tn_wide_from();
        if (O > 128 && O <= 192) return wide ? EEG_TNW(6, 16) : EEG_TN(6, 16);
        if (O > 64 && O <= 128) return wide ? EEG_TNW(4, 32) : EEG_TN(4, 32);
        return wide ? EEG_TNW(2, 32) : EEG_TN(2, 32);
#undef EEG_TNW
#undef EEG_TN
    }
    if (bt.T > 0) return fail("gemm_tn: a batch-major segment needs the LDS-DMA kernel (O=%d)", O);
    if (O <= 32) return run_tn<1>(segs, nseg, F, R, dY, ldy, ycol0, O, partial, nsplit, rows_per_split, st, tag);
    if (O <= 64) return run_tn<2>(segs, nseg, F, R, dY, ldy, ycol0, O, partial, nsplit, rows_per_split, st, tag);
    if (O <= 128) return run_tn<4>(segs, nseg, F, R, dY, ldy, ycol0, O, partial, nsplit, rows_per_split, st, tag);
    if (O <= 192) return run_tn<6>(segs, nseg, F, R, dY, ldy, ycol0, O, partial, nsplit, rows_per_split, st, tag);
    return fail("gemm_tn: O=%d unsupported", O);
}

// the streaming kernel applies (and with it the batch-major input option of eeg_dcrnn_layer_fwd)
bool diffuse_streams(int p_batched, int S, int B, int N, int F) { return N == 19 && F / 4 <= 128 && S / (p_batched ? B : 1) >= 4; }

// x_bt: X is batch-major (B, S/B, N, F) and xcopy gets its time-major copy (streaming kernel only)
int diffuse_fwd(const float* X, const float* P, int p_batched, int S, int B, int N, int F, int M, float* planes,
                hipStream_t st, size_t plane_stride = 0, int x_bt = 0, float* xcopy = nullptr) {
    if (plane_stride == 0) plane_stride = (size_t)S * N * F;
    if (M == 1 && xcopy == nullptr) return 0;         // max_diffusion_step = 0: no planes (a requested copy still runs below)
    if (x_bt && !diffuse_streams(p_batched, S, B, N, F)) return fail("diffuse_fwd: batch-major input needs the streaming kernel");
    // the EEG montage: streaming kernel (no LDS); launches with < 4 samples per graph (decoder steps) leave most of its
    // lanes idle and are faster through the LDS/MFMA kernel below
    if (diffuse_streams(p_batched, S, B, N, F)) {
        const int F4 = F / 4, sB = p_batched ? B : 1, T = S / sB;
        int threads = 256;                        // few samples per graph: narrower workgroups
        while (threads > 64 && (threads / 2) / F4 >= T && (threads / 2) >= F4) threads /= 2;
        const int SPW = threads / F4;
        int ny = ceil_div(T, SPW);
        // ~4 workgroups per CU in total, each walking >= 2 passes of consecutive samples: measured on MI355X (cfg2, F=100,
        // profiles/r02_b_diffuse_sweep.txt) 64 us against 75 us with 6 single-pass workgroups per CU and 68 / 85 us with 2 / 1
        // (fewer, longer sequential streams suit the DRAM pages better than more parallelism); prefetching the next
        // sample's rows ahead of the stores was slower (71 us).  dev knob 5 overrides the target.
        const int want = ceil_div(g_tune[5] > 0 ? g_tune[5] : 1024, sB);
        if (ny > want) ny = want;
        if (ny < 1) ny = 1;
        EEG_LAUNCH_P("diffuse_fwd", diffuse_fwd_stream_kernel<19>, dim3(sB, ny), dim3(threads), 0, st, X, P, p_batched, S, B, F, M, planes, plane_stride, x_bt, xcopy);
        return check_launch("diffuse_fwd");
    }
    // LDS kernel: rows wider than one tile (LDS budget / prefetch registers) go in column chunks, one launch each
    const int NR = round_up(N, 4);
    int Fc = F;
    auto lds_of = [&](int fc) { return ((size_t)(M - 1) * kPFloats + (size_t)NR * lds_stride(M * round_up(fc, 16))) * sizeof(float); };
    while (Fc > 16 && (lds_of(Fc) > 96 * 1024 || N * (Fc / 4) > 256 * kDiffPrefetch)) Fc = round_up(ceil_div(Fc, 2), 16);
    const size_t lds = lds_of(Fc);
    if (lds > kMaxLdsBytes || N * (Fc / 4) > 256 * kDiffPrefetch) return fail("diffuse_fwd: N=%d M=%d does not fit the LDS tile", N, M);
    EEG_SET_MAX_LDS(diffuse_fwd_kernel, lds);
    dim3 grid;
    if (p_batched) {
        const int T = S / B;
        int tg = ceil_div(2048, B);
        if (tg > T) tg = T;
        grid = dim3(B, tg < 1 ? 1 : tg);
    } else {
        grid = dim3(S < 2048 ? S : 2048, 1);
    }
    for (int c0 = 0; c0 < F; c0 += Fc) {
        EEG_LAUNCH_P("diffuse_fwd", diffuse_fwd_kernel, grid, dim3(256), lds, st, X, P, p_batched, S, B, N, F, M, planes, plane_stride,
                     c0, F - c0 < Fc ? F - c0 : Fc);
        if (check_launch("diffuse_fwd")) return 1;
    }
    return 0;
}
int diffuse_adj(const float* Z, const float* P, int p_batched, int S, int B, int N, int F, int M, float* dX,
                hipStream_t st, const float* add = nullptr) {
    if (N == 19 && F / 4 <= 128 && g_tune[9] == 0 && (double)S * N * M * F < 1.7e10 && S / (p_batched ? B : 1) >= 4) {   // the EEG montage: streaming kernel (no LDS); 32-bit float4 offsets
        const int F4 = F / 4, sB = p_batched ? B : 1, T = S / sB;
        int threads = 256;
        while (threads > 64 && (threads / 2) / F4 >= T && (threads / 2) >= F4) threads /= 2;
        const int SPW = threads / F4;
        int ny = ceil_div(T, SPW);
        const int want = ceil_div(g_tune[6] > 0 ? g_tune[6] : 4096, sB);   // dev knob 6 (512 .. 4096 measured alike)
        if (ny > want) ny = want;
        if (ny < 1) ny = 1;
        // round 5: Z walked in storage order (all hop slots of a node row together) where the hop count has an instantiation
        if (g_tune[18] == 0 && M == 3) {
            EEG_LAUNCH_P("diffuse_adj", (diffuse_adj_rows_kernel<19, 3>), dim3(sB, ny), dim3(threads), 0, st, Z, P, p_batched, S, B, F, add, dX);
        } else if (g_tune[18] == 0 && M == 5) {
            EEG_LAUNCH_P("diffuse_adj", (diffuse_adj_rows_kernel<19, 5>), dim3(sB, ny), dim3(threads), 0, st, Z, P, p_batched, S, B, F, add, dX);
        } else {
            EEG_LAUNCH_P("diffuse_adj", diffuse_adj_stream_kernel<19>, dim3(sB, ny), dim3(threads), 0, st, Z, P, p_batched, S, B, F, M, add, dX);
        }
        return check_launch("diffuse_adj");
    }
    const int NR = round_up(N, 4);
    int Fc = F;
    auto lds_of = [&](int fc) { return ((size_t)(M - 1) * kPFloats + (size_t)NR * lds_stride(M * round_up(fc, 16))) * sizeof(float); };
    while (Fc > 16 && lds_of(Fc) > 96 * 1024) Fc = round_up(ceil_div(Fc, 2), 16);
    const size_t lds = lds_of(Fc);
    if (lds > kMaxLdsBytes) return fail("diffuse_adj: N=%d M=%d does not fit the LDS tile", N, M);
    EEG_SET_MAX_LDS(diffuse_adj_kernel, lds);
    const int grid = S < 2048 ? S : 2048;
    for (int c0 = 0; c0 < F; c0 += Fc) {
        EEG_LAUNCH_P("diffuse_adj", diffuse_adj_kernel, dim3(grid), dim3(256), lds, st, Z, P, p_batched, S, B, N, F, M, add, dX,
                     c0, F - c0 < Fc ? F - c0 : Fc);
        if (check_launch("diffuse_adj")) return 1;
    }
    return 0;
}

// the two-wave recurrent kernels exist in a SPEC instantiation (spectral form: seq_launch.h) for exactly these calls -- mirrors the
// selection inside seq_inst.cpp (64 units, at most 20 nodes, 2 or 3 hop matrices, two-wave variant on, 2 GB buffer descriptors)
bool seq2_spec_ok(int H, int M, int N, int T, int B, int variant, int Sp, int SpE) {
    if (H != 64 || N < 16 || N > 20 || M < 2 || M > 3 || variant != 1 || (phase_probe_armed() && M != 3)) return false;   // (probe instantiations: M = 3)
    return (double)T * B * N * 3 * H * sizeof(float) < 2147483648.0 && (double)N * Sp * 3 * H * sizeof(float) < 2147483648.0 &&
           (double)N * SpE * H * sizeof(float) < 2147483648.0;
}

int seq_fwd(int H, int M, const SeqFwdArgs& a, hipStream_t st) {
    int rc = H == 16 ? launch_seq_fwd_h16(M, a, st) : H == 32 ? launch_seq_fwd_h32(M, a, st) : launch_seq_fwd_h64(M, a, st);
    if (rc == 1) return fail("seq_fwd: no kernel for H=%d M=%d", H, M);
    if (rc == 2) return fail("seq_fwd: kernel launch failed (H=%d M=%d)", H, M);
    if (rc == 3) return fail("seq_fwd: H=%d M=%d N=%d exceeds the LDS of a CU", H, M, a.N);
    return 0;
}
// BPTT with more clips than 1.5 x the CUs at hop counts the two-wave kernel does not cover: two streamed-weight
// workgroups per CU (kernels_seq_stream.h; dev knob 3: 1 = wherever the kernel exists, 2 = never)
bool seq_stream_wanted(int H, int M, int B, int N) {
    if (H != 64 || N > kDecRows || M > 5 || g_tune[3] == 2) return false;
    return g_tune[3] == 1 || (M >= 4 && B >= 384);
}
int seq_bwd(int H, int M, const SeqBwdArgs& a, hipStream_t st) {
    if (a.variant != 0 && !phase_probe_armed() && seq_stream_wanted(H, M, a.B, a.N)) {   // (the streamed kernel takes no clock samples either)
        const int rs = launch_seq_bwd_stream(M, a, st);
        if (rs == 0) return 0;
        if (rs == 2) return fail("seq_bwd: streamed kernel launch failed (M=%d)", M);
    }
    int rc = H == 16 ? launch_seq_bwd_h16(M, a, st) : H == 32 ? launch_seq_bwd_h32(M, a, st) : launch_seq_bwd_h64(M, a, st);
    if (rc == 1) return fail("seq_bwd: no kernel for H=%d M=%d", H, M);
    if (rc == 2) return fail("seq_bwd: kernel launch failed (H=%d M=%d)", H, M);
    if (rc == 3) return fail("seq_bwd: H=%d M=%d N=%d exceeds the LDS of a CU", H, M, a.N);
    return 0;
}

struct BwdWs {
    size_t dxw, dbias, hplanes, rhplanes, partial, part_g, part_c, z, total;   // partial = x-part region; part_g / part_c follow it
    int nsplit_x, rps_x, nsplit_hg, rps_hg, nsplit_hc, rps_hc;
    TnqPlan qx, qg, qc;        // round-3 TN kernel where it covers the shape (ok = 1): its split plan replaces tn_split's
    // spectral form (d->spectral): dyh = U^T dXW (N, Sp, 3H); hplanes / rhplanes = U^T h_{t-1}, U^T (r*h_{t-1}) (N, Sp, H); partial /
    // part_g / part_c = the grouped TNs' [N*spg][K][O]; z = dXh (N, Sp, Fin)
    size_t dyh;
    TngPlan gx, gh;
    TnfPlan gf;                // the three problems in one pass (kernels_gemm_f.h); ok: gx.spg = gh.spg = gf.spg
};
// dev knob 2 bit 1 = 2: the round-2 TN kernels everywhere
TnqPlan tn_plan_q(int nseg, int F, int R, int O, bool bt) {
    if ((g_tune[2] & 2) != 0 || g_tune[1] != 0 || R < q_min_rows()) return TnqPlan{};
    return tnq_plan(nseg, F, R, O, bt, g_tune[16] > 0 ? g_tune[16] / 2 : num_cus());    // dev knob 16: target workgroups of the whole-block TN GEMM
}
BwdWs bwd_ws(const eeg_layer_dims* d, int need_dx) {
    BwdWs w;
    const size_t R = (size_t)d->T * d->B * d->N;
    const bool spec = d->spectral != nullptr;
    const size_t Rp = (size_t)d->N * spec_rows(d->T * d->B);
    size_t o = 0;
    w.dxw = o;      o += R * 3 * d->H;
    w.dbias = o;    o += round_up(d->B * 3 * d->H, 64);
    w.hplanes = o;  o += spec ? Rp * d->H : (size_t)(d->M - 1) * R * d->H;
    w.rhplanes = o; o += spec ? Rp * d->H : (size_t)(d->M - 1) * R * d->H;
    w.dyh = o;      o += spec ? Rp * 3 * d->H : 0;
    w.gx = spec ? tng_plan(d->Fin, spec_rows(d->T * d->B), d->N, num_cus()) : TngPlan{};
    w.gh = spec ? tng_plan(d->H, spec_rows(d->T * d->B), d->N, num_cus()) : TngPlan{};
    // dev knob 23 = 1: the three separate grouped launches
    w.gf = spec && g_tune[23] == 0 ? tnf_plan(d->Fin, d->H, spec_rows(d->T * d->B), d->N, num_cus()) : TnfPlan{};
    if (w.gf.ok) { w.gx.spg = w.gh.spg = w.gf.spg; w.gx.rps = w.gh.rps = w.gf.rps; }
    w.nsplit_x = tn_split(d->M, d->Fin, (int)R, 3 * d->H, &w.rps_x);
    w.nsplit_hg = tn_split(d->M, d->H, (int)R, 2 * d->H, &w.rps_hg);
    w.nsplit_hc = tn_split(d->M, d->H, (int)R, d->H, &w.rps_hc);
    w.qx = tn_plan_q(d->M, d->Fin, (int)R, 3 * d->H, d->x_batch_major != 0);
    w.qg = tn_plan_q(d->M, d->H, (int)R, 2 * d->H, false);
    w.qc = tn_plan_q(d->M, d->H, (int)R, d->H, false);
    if (w.qx.ok) { w.nsplit_x = w.qx.nsplit; w.rps_x = w.qx.rps; }
    if (w.qg.ok) { w.nsplit_hg = w.qg.nsplit; w.rps_hg = w.qg.rps; }
    if (w.qc.ok) { w.nsplit_hc = w.qc.nsplit; w.rps_hc = w.qc.rps; }
    size_t px = (size_t)w.nsplit_x * d->M * d->Fin * 3 * d->H;
    size_t pg = (size_t)w.nsplit_hg * d->M * d->H * 2 * d->H;
    size_t pc = (size_t)w.nsplit_hc * d->M * d->H * d->H;
    if (spec) {
        px = (size_t)d->N * w.gx.spg * d->Fin * 3 * d->H;
        pg = (size_t)d->N * w.gh.spg * d->H * 2 * d->H;
        pc = (size_t)d->N * w.gh.spg * d->H * d->H;
    }
    w.partial = o;  o += (px + 63) / 64 * 64;      // one region per GEMM: the three are reduced by one launch
    w.part_g = o;   o += (pg + 63) / 64 * 64;
    w.part_c = o;   o += (pc + 63) / 64 * 64;
    w.z = o;        o += need_dx ? (spec ? Rp * d->Fin : R * d->M * d->Fin) : 0;
    w.total = o;
    return w;
}


// Hoisted weight gradients of one cell over all R = T*B*N rows (split-K GEMMs with fixed-order
// reduction): x-part [X | P_m X]^T [dR|dU|dC], h-part of the gate hops(h_{t-1})^T [dR|dU] and of the
// candidate hops(r*h_{t-1})^T dC.  accumulate = add into dWg/dWc (a cell shared by several layers).
// hpl_in / rpl_in: hop planes of h_{t-1} / r*h_{t-1} kept by the forward kernel (NULL: recomputed into hpl / rpl).
// x_stride / h_stride: floats between two hop planes of `planes` / of hpl_in, rpl_in.
int cell_weight_grads(const eeg_layer_dims* d, const float* X, const float* planes, size_t x_stride, const float* Hprev,
                      const float* RHs, const float* dXW, const float* P, const float* hpl_in, const float* rpl_in,
                      size_t h_stride, float* hpl_ws, float* rpl_ws, float* part, const BwdWs& w, bool accumulate,
                      float* dWg, float* dWc, hipStream_t st, BtMap bt = BtMap(), const float* bias_part = nullptr,
                      float* dbg = nullptr, float* dbc = nullptr) {
    const int S = d->T * d->B, R = S * d->N, H = d->H, M = d->M, Fin = d->Fin, N = d->N;
    const int acc = accumulate ? 8 : 0;
    SegPtrs sx;
    for (int m = 0; m < kMaxM; ++m) sx.p[m] = m == 0 ? X : (m < M ? planes + (size_t)(m - 1) * x_stride : nullptr);
    float* part_g = part + (w.part_g - w.partial);
    float* part_c = part + (w.part_c - w.partial);
    if (gemm_tn(sx, M, Fin, R, dXW, 3 * H, 0, 3 * H, part, w.nsplit_x, w.rps_x, st, "gemm_tn_x", bt, &w.qx)) return 1;
    //   h-part of the gate: hops(h_{t-1})^T [dR|dU]
    const float* hpl = hpl_in;
    size_t hs = h_stride;
    if (hpl == nullptr) {
        if (diffuse_fwd(Hprev, P, d->p_batched, S, d->B, N, H, M, hpl_ws, st)) return 1;
        hpl = hpl_ws;
        hs = (size_t)R * H;
    }
    SegPtrs sh;
    for (int m = 0; m < kMaxM; ++m) sh.p[m] = m == 0 ? Hprev : (m < M ? hpl + (size_t)(m - 1) * hs : nullptr);
    //   h-part of the candidate: hops(r*h_{t-1})^T dC
    const float* rpl = rpl_in;
    size_t rs = h_stride;
    if (rpl == nullptr) {
        if (diffuse_fwd(RHs, P, d->p_batched, S, d->B, N, H, M, rpl_ws, st)) return 1;
        rpl = rpl_ws;
        rs = (size_t)R * H;
    }
    SegPtrs sr;
    for (int m = 0; m < kMaxM; ++m) sr.p[m] = m == 0 ? RHs : (m < M ? rpl + (size_t)(m - 1) * rs : nullptr);
    // Round 5: where the whole-block TN kernel covers both h-part problems with the same plan (same K, row splits and dY rows; only
    // the column count differs: 2H vs H) they go out as ONE launch whose workgroups alternate between the two -- the 64-column
    // problem waits for its operands (0.61 of the MFMA peak alone), the 128-column one for the matrix pipe (0.77): side by side on
    // the two workgroup slots of a CU they take 0.366 instead of 0.390 ms per cfg2 step (role `gemm_tn_h`; dev knob 19 = 1: one by one)
    int pair = -1;
    if (g_tune[19] == 0) {
        pair = launch_tnq_pair(w.qg, w.qc, sh, sr, M, H, R, dXW, 3 * H, 0, 2 * H, part_g, 2 * H, H, part_c, st, "gemm_tn_h");
        if (pair > 0) return fail("gemm_tnq_pair: launch failed");
    }
    if (pair < 0) {
        if (gemm_tn(sh, M, H, R, dXW, 3 * H, 0, 2 * H, part_g, w.nsplit_hg, w.rps_hg, st, "gemm_tn_hg", BtMap(), &w.qg)) return 1;
        if (gemm_tn(sr, M, H, R, dXW, 3 * H, 2 * H, H, part_c, w.nsplit_hc, w.rps_hc, st, "gemm_tn_hc", BtMap(), &w.qc)) return 1;
    }
    //   one fixed-order reduction of the three sets of split-K partials into the reference's gradient layout
    ReduceJobs jobs;
    const int Ks[3] = {M * Fin, M * H, M * H}, Os[3] = {3 * H, 2 * H, H}, ns[3] = {w.nsplit_x, w.nsplit_hg, w.nsplit_hc};
    const float* parts[3] = {part, part_g, part_c};
    int nblocks = 0;
    for (int j = 0; j < 3; ++j) {
        jobs.part[j] = parts[j]; jobs.nsplit[j] = ns[j]; jobs.K[j] = Ks[j]; jobs.O[j] = Os[j];
        jobs.nblocks[j] = ceil_div(Ks[j] * Os[j], 64);
        nblocks += jobs.nblocks[j];
    }
    // the cell's bias gradients (per-clip partials of the BPTT kernel -> dbg, dbc) as a fourth job of the same launch
    jobs.bias_part = bias_part; jobs.bias_B = d->B; jobs.dbg = dbg; jobs.dbc = dbc;
    if (bias_part != nullptr) nblocks += ceil_div(3 * H, 16);
    EEG_LAUNCH_P("reduce_unpack", reduce_unpack3_kernel, dim3(nblocks), dim3(256), 256 * sizeof(float4), st, jobs, acc, Fin, H, M, dWg, dWc);
    return check_launch("reduce_unpack");
}

// The same weight gradients in the eigenbasis of a shared symmetric support (spec_common.h): every hoisted contraction runs per
// graph frequency i over K = Fin (x-part) / K = H (h-parts) instead of M * Fin / M * H, and the fold dW_m = sum_i T_m(lam_i) dWt_i
// rides in the reduction launch.  Xh = U^T X (N, Sp, Fin; groups x_gs floats apart), dYh = U^T dXW (N, Sp, 3H), hh = U^T h_{t-1}
// (groups hh_gs floats apart, 0 = contiguous), rhh = U^T (r*h_{t-1}) (N, Sp, H); all rows time-major.
size_t xs_spec(const eeg_layer_dims* d) { return d->x_plane_stride > 0 ? (size_t)d->x_plane_stride : 0; }
int cell_weight_grads_spectral(const eeg_layer_dims* d, const float* Xh, size_t x_gs, const float* hh, size_t hh_gs, const float* rhh,
                               const float* dYh, float* part, const BwdWs& w, float* dWg, float* dWc, hipStream_t st,
                               const float* bias_part, float* dbg, float* dbc) {
    const int S = d->T * d->B, H = d->H, M = d->M, Fin = d->Fin, N = d->N, Sp = spec_rows(S);
    float* part_g = part + (w.part_g - w.partial);
    float* part_c = part + (w.part_c - w.partial);
    if (w.gf.ok) {
        if (launch_tnf(w.gf, Xh, x_gs, Fin, hh, hh_gs, rhh, dYh, Sp, N, part, part_g, part_c, st, "gemm_tn_f")) return fail("gemm_tnf: launch failed");
    } else {
        if (launch_tng(w.gx, Xh, Fin, Sp, N, dYh, part, st, "gemm_tn_x", x_gs)) return fail("gemm_tng: launch failed");
        if (launch_tng_pair(w.gh, hh, rhh, Sp, N, dYh, part_g, part_c, st, "gemm_tn_h", hh_gs)) return fail("gemm_tng_pair: launch failed");
    }
    ReduceJobs jobs{};
    SpecFoldJobs sj{};
    sj.basis = d->spectral; sj.N = N;
    sj.j[0] = SpecFoldJob{part, w.gx.spg, Fin, 3 * H, ceil_div(Fin * 3 * H, 64)};
    sj.j[1] = SpecFoldJob{part_g, w.gh.spg, H, 2 * H, ceil_div(H * 2 * H, 64)};
    sj.j[2] = SpecFoldJob{part_c, w.gh.spg, H, H, ceil_div(H * H, 64)};
    int nblocks = sj.j[0].nblocks + sj.j[1].nblocks + sj.j[2].nblocks;
    jobs.bias_part = bias_part; jobs.bias_B = d->B; jobs.dbg = dbg; jobs.dbc = dbc;
    if (bias_part != nullptr) nblocks += ceil_div(3 * H, 16);
    const int ns_max = N * (w.gx.spg > w.gh.spg ? w.gx.spg : w.gh.spg);
    const size_t fold_lds = spec_fold_lds_bytes(M, ns_max);
    EEG_SET_MAX_LDS(reduce_unpack3s_kernel, fold_lds);
    EEG_LAUNCH_P("reduce_unpack", reduce_unpack3s_kernel, dim3(nblocks), dim3(256), fold_lds, st, jobs, sj, 0, Fin, H, M, dWg, dWc);
    return check_launch("reduce_unpack (spectral)");
}

// out0[c] (c < split) / out1[c - split] = sum_r A[r][c]: two fixed-order stages.
int colsum_rows_per_chunk(int R) { int rpc = ceil_div(R, 512); return rpc < 64 ? 64 : rpc; }
size_t colsum_ws(int R, int C) { return (size_t)ceil_div(R, colsum_rows_per_chunk(R)) * C; }
int colsum(const float* A, int R, int C, int split, float* ws, float* out0, float* out1, hipStream_t st) {
    const int rpc = colsum_rows_per_chunk(R), nchunk = ceil_div(R, rpc);
    if (C % 4 == 0 && C <= 1024) {
        const int nrs = 256 / (C / 4);
        EEG_LAUNCH_P("reduce_bias", colsum_partial4_kernel, dim3(nchunk), dim3(256), (size_t)nrs * C * sizeof(float), st, A, R, C, rpc, ws);
    } else {
        EEG_LAUNCH_P("reduce_bias", colsum_partial_kernel, dim3(nchunk, ceil_div(C, 128)), dim3(256), 256 * sizeof(float), st, A, R, C, rpc, ws);
    }
    if (check_launch("colsum_partial")) return 1;
    EEG_LAUNCH_P("reduce_bias", colsum_final_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, st, ws, nchunk, C, split, out0, out1);
    return check_launch("colsum_final");
}

// ---- decoder (model.py:112-204): buffer carving shared by forward and backward -------------------
struct DecLayout {
    // saved (forward -> backward)
    size_t xin, proj_pack, proj_bias, hd, saved_total;
    size_t planes[8], hext[8], rs[8], us[8], cs[8], rhs[8], hpl[8], rpl[8];
    // backward workspace
    size_t dxw[8], dbias[8], dotot, da, dhn, z, partial, projt_pack, colsum, bwd_total;
    int nsplit_p, rps_p;
    BwdWs lw[8];
    eeg_layer_dims ld[8];
};
size_t align64(size_t v) { return (v + 63) / 64 * 64; }
DecLayout dec_layout(const eeg_decoder_dims* d) {
    DecLayout y;
    const size_t state = (size_t)d->B * d->N * d->H, R = (size_t)d->T * d->B * d->N;
    const int nct_o = ceil_div(d->Dout, 16), nct_h = ceil_div(d->H, 16);
    size_t o = 0;
    y.xin = o;       o += align64(R * d->Dout);
    y.proj_pack = o; o += align64((size_t)(d->H / 4) * nct_o * 64);
    y.proj_bias = o; o += align64((size_t)nct_o * 16);
    for (int l = 0; l < d->L; ++l) {
        const int fin = l == 0 ? d->Dout : d->H;
        y.ld[l] = eeg_layer_dims{d->T, d->B, d->N, d->H, fin, d->M, d->act, d->p_batched, 0, 0, 0};
        y.planes[l] = o; if (l == 0) o += align64((size_t)(d->M - 1) * R * fin);   // layers >= 1: the hpl of the layer below
        y.hext[l] = o;   o += align64((size_t)(d->T + 1) * state);
        y.rs[l] = o;     o += align64((size_t)d->T * state);
        y.us[l] = o;     o += align64((size_t)d->T * state);
        y.cs[l] = o;     o += align64((size_t)d->T * state);
        y.rhs[l] = o;    o += align64((size_t)d->T * state);
        // hop planes of h_{t-1} (slots 0..T; slot t+1 = hops of this layer's output at step t = the next layer's
        // input planes) and of r*h_{t-1}: by-products of the forward kernel, A operands of the dW GEMMs
        y.hpl[l] = o;    o += align64((size_t)(d->M - 1) * (d->T + 1) * state);
        y.rpl[l] = o;    o += align64((size_t)(d->M - 1) * (d->T + 1) * state);
    }
    // dropout in front of the projection: the dropped top-layer outputs (T,B,N,H) (A operand of dW_p); without dropout they
    // ARE hext[L-1] slots 1..T
    y.hd = d->dropout_p > 0.f ? o : y.hext[d->L - 1] + state;
    if (d->dropout_p > 0.f) o += align64((size_t)d->T * state);
    y.saved_total = o;
    o = 0;
    size_t part = 0;
    for (int l = 0; l < d->L; ++l) {
        y.lw[l] = bwd_ws(&y.ld[l], 0);
        y.dxw[l] = o; o += align64(R * 3 * d->H);
        const size_t pl = y.lw[l].total - y.lw[l].partial;      // partial is the last region when need_dx = 0
        part = pl > part ? pl : part;
    }
    for (int l = 0; l < d->L; ++l) { y.dbias[l] = o; o += (size_t)d->T * d->B * 3 * d->H; }   // contiguous: shared layers reduce at once
    o = align64(o);
    y.dotot = o;  o += align64(R * d->Dout);
    y.da = o;     o += align64(state);
    y.dhn = o;    o += align64(2 * (size_t)d->L * state);
    y.z = o;      o += align64((size_t)d->B * d->N * d->M * (d->Dout > d->H ? d->Dout : d->H));
    y.nsplit_p = tn_split(1, d->Dout, (int)R, d->H, &y.rps_p);
    const size_t pp = (size_t)y.nsplit_p * d->Dout * d->H;
    part = pp > part ? pp : part;
    y.partial = o; o += align64(part);
    y.projt_pack = o; o += align64((size_t)(d->Dout / 4) * nct_h * 64);
    const size_t c1 = colsum_ws((int)R, d->Dout), c2 = colsum_ws(d->L * d->T * d->B, 3 * d->H);
    y.colsum = o; o += align64(c1 > c2 ? c1 : c2);
    y.bwd_total = o;
    return y;
}
// nn.Dropout's own argument check (torch: "dropout probability has to be between 0 and 1, but got ...")
int check_dropout_p(const char* who, float p) {
    if (!(p >= 0.f && p <= 1.f)) return fail("%s: dropout probability has to be between 0 and 1, but got %g", who, p);
    return 0;
}
int check_decoder_dims(const eeg_decoder_dims* d) {
    if (d->L < 1 || d->L > 8) return fail("decoder: num_rnn_layers=%d unsupported (1..8)", d->L);
    if (d->T < 1 || d->B < 1) return fail("decoder: empty sequence/batch (T=%d, B=%d)", d->T, d->B);
    if (check_dropout_p("decoder", d->dropout_p)) return 1;
    if (check_dims(d->N, d->H, d->Dout, d->M)) return 1;
    return check_dims(d->N, d->H, d->H, d->M);
}
int copy_floats(float* dst, const float* src, size_t n, hipStream_t st) {
    return platform_copy_floats(dst, src, n, st) ? 0 : fail("device copy failed");
}

}  // namespace

extern "C" {

const char* eeg_dcrnn_last_error(void) { return g_err; }
int eeg_dcrnn_abi_version(void) { return 5; }
int eeg_dcrnn_is_device_build(void) { return kPlatformIsDevice; }
#if defined(EEG_DEV)
int eeg_dcrnn_set_tuning(int key, int value) {
    if (key < 0 || key >= 24) return fail("set_tuning: key %d out of range", key);
    g_tune[key] = value;
    return 0;
}
int eeg_dcrnn_set_seq_probe(int64_t* probe) {
    g_seq_probe = reinterpret_cast<long long*>(probe);
    return 0;
}
#endif
int eeg_dcrnn_prof_enable(int on) {
    eeg::prof_enable(on != 0);
    return 0;
}
int eeg_dcrnn_prof_report(char* buf, size_t cap) {
    if (buf == nullptr || cap == 0) return fail("prof_report: empty buffer");
    buf[0] = 0;
    const size_t need = eeg::prof_report(buf, cap);
    if (need != 0) return fail("prof_report: buffer too small (%zu needed)", need);
    return 0;
}
namespace {
// Every SIMD of the chip streams fp32 MFMAs (two accumulator chains per wave = the issue rate) for `ticks` of the chip-wide 100 MHz
// counter; the shader-clock cycles that pass during the SECOND half are summed over the workgroups: out[0] += cycles, out[1] += ticks.
__global__ __launch_bounds__(256) void clock_probe_kernel(long long ticks, long long* __restrict__ out) {
    const float a = 0.001f * (threadIdx.x & 63), b = 1.f + (threadIdx.x & 63);
    f32x4 p = {0.f, 0.f, 0.f, 0.f}, q = p;
    const long long r0 = realtime_now();
    long long r1 = r0, rh = 0, ch = 0;
    for (int i = 0; i < (1 << 20) && r1 - r0 < ticks; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { p = mfma16(a, b, p); q = mfma16(b, a, q); }
        r1 = realtime_now();
        if (rh == 0 && r1 - r0 >= ticks / 2) { rh = r1; ch = cycle_now(); }
    }
    const long long c1 = cycle_now();
    if (p[0] + q[0] == 12345.f) out[2] = 1;                       // (keeps the MFMAs)
    if (threadIdx.x == 0 && rh != 0) {
        atomic_add_u64(reinterpret_cast<unsigned long long*>(out), (unsigned long long)(c1 - ch));
        atomic_add_u64(reinterpret_cast<unsigned long long*>(out) + 1, (unsigned long long)(r1 - rh));
    }
}
}  // namespace
int eeg_dcrnn_prof_clock_samples(int64_t* buf4) {
    g_clock_samples.store(reinterpret_cast<long long*>(buf4), std::memory_order_release);
    return 0;
}
int eeg_dcrnn_prof_clock_probe(int64_t* out2, void* stream) {
    if (out2 == nullptr) return fail("prof_clock_probe: null output");
    // 200 us of full-chip fp32-MFMA load; the caller zeroes out2 (3 x int64) first
    EEG_LAUNCH(clock_probe_kernel, dim3(platform_num_cus()), dim3(256), 0, S_(stream), (long long)20000, reinterpret_cast<long long*>(out2));
    return check_launch("clock_probe");
}
int eeg_dcrnn_supported(int N, int H, int Fin, int M) { return check_dims(N, H, Fin, M) == 0 ? 1 : 0; }
int eeg_dcrnn_zero(void* p, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    if (p == nullptr) return fail("zero: null pointer");
    if (reinterpret_cast<uintptr_t>(p) % 16 != 0)                       // (torch allocations are 256-byte aligned; views may not be)
        return hipMemsetAsync(p, 0, bytes, S_(stream)) == hipSuccess ? 0 : fail("zero: memset failed");
    const size_t n16 = bytes / 16;
    const int blocks = (int)(n16 / 256 < 1 ? 1 : (n16 / 256 > 1024 ? 1024 : n16 / 256));
    EEG_LAUNCH_P("zero", zero_kernel, dim3(blocks), dim3(256), 0, S_(stream), reinterpret_cast<float4*>(p), n16,
                 reinterpret_cast<unsigned char*>(p) + 16 * n16, (int)(bytes - 16 * n16));
    return check_launch("zero");
}

int eeg_dcrnn_hop_polys(const float* const* supports, int n_supports, int n_graphs, int N, int K, float* P_out,
                        void* stream) {
    if (n_supports < 1 || n_supports > 4) return fail("hop_polys: n_supports=%d unsupported (1..4)", n_supports);
    if (N < 1 || N > kMaxNodes) return fail("hop_polys: num_nodes=%d unsupported", N);
    if (K < 0) return fail("hop_polys: max_diffusion_step=%d unsupported (>= 0)", K);
    if (n_graphs < 1) return fail("hop_polys: no graphs (n_graphs=%d)", n_graphs);
    if (K == 0) return 0;                            // cell.py:80-81: no hop matrices beyond the identity, P_out is empty
    if (n_supports * K + 1 > kMaxM) return fail("hop_polys: %d supports x K=%d exceeds %d hop matrices", n_supports, K, kMaxM);
    SupPtrs sp;
    for (int i = 0; i < 4; ++i) sp.p[i] = i < n_supports ? supports[i] : nullptr;
    const size_t lds = 4 * kMaxNodes * kMaxNodes * sizeof(float);
    EEG_LAUNCH_P("hop_polys", hop_polys_kernel, dim3(n_graphs), dim3(round_up(N * N, 64)), lds, S_(stream), sp, n_supports, N, K, P_out);
    return check_launch("hop_polys");
}

size_t eeg_dcrnn_pack_floats(int Fin, int H, int M) { return make_cell_pack(Fin, H, M).total; }

int eeg_dcrnn_pack_cell(const float* Wg, const float* bg, const float* Wc, const float* bc, int Fin, int H, int M,
                        float* pack, void* stream) {
    if (!h_supported(H)) return fail("pack_cell: rnn_units=%d unsupported", H);
    if (Fin < 4 || Fin % 4 != 0) return fail("pack_cell: input dim %d must be a positive multiple of 4", Fin);
    if (!m_supported(M)) return fail("pack_cell: num hop matrices M=%d unsupported (1,2,3,4,5,7)", M);
    CellPack p = make_cell_pack(Fin, H, M);
    EEG_LAUNCH_P("pack_cell", pack_cell_kernel, dim3(512), dim3(256), 0, S_(stream), Wg, bg, Wc, bc, pack, p);
    return check_launch("pack_cell");
}
size_t eeg_dcrnn_pack3_halves(int Fin, int H, int M) { return pack3_supported(Fin, H, M) ? make_pack3(Fin, H, M).total : 0; }
int eeg_dcrnn_pack_cell_bf16x3(const float* Wg, const float* Wc, int Fin, int H, int M, uint16_t* pack3, void* stream) {
    if (!pack3_supported(Fin, H, M)) return fail("pack_cell_bf16x3: rnn_units=%d, input_dim=%d, %d hop matrices: the bf16 split exists for 64 units", H, Fin, M);
    if (Wg == nullptr || Wc == nullptr || pack3 == nullptr) return fail("pack_cell_bf16x3: null pointer");
    EEG_LAUNCH_P("pack_cell", pack_cell_bf3_kernel, dim3(512), dim3(256), 0, S_(stream), Wg, Wc, Fin, H, M, pack3);
    return check_launch("pack_cell_bf3");
}

int eeg_dcrnn_diffuse_fwd(const float* X, const float* P, int p_batched, int S, int B, int N, int F, int M,
                          float* planes, void* stream) {
    if (N < 1 || N > kMaxNodes || F < 4 || F % 4 != 0 || M < 1 || M > kMaxM) return fail("diffuse_fwd: bad dims N=%d F=%d M=%d", N, F, M);
    if (S < 1 || B < 1) return fail("diffuse_fwd: empty input (S=%d, B=%d)", S, B);
    return diffuse_fwd(X, P, p_batched, S, B, N, F, M, planes, S_(stream));
}
int eeg_dcrnn_diffuse_adj(const float* Z, const float* P, int p_batched, int S, int B, int N, int F, int M,
                          float* dX, void* stream) {
    if (N < 1 || N > kMaxNodes || F < 4 || F % 4 != 0 || M < 1 || M > kMaxM) return fail("diffuse_adj: bad dims N=%d F=%d M=%d", N, F, M);
    if (S < 1 || B < 1) return fail("diffuse_adj: empty input (S=%d, B=%d)", S, B);
    return diffuse_adj(Z, P, p_batched, S, B, N, F, M, dX, S_(stream));
}

// Size queries never fail and never fault: dims the compute entry would refuse (empty batch / sequence, zero widths) size to 0.
static bool layer_dims_positive(const eeg_layer_dims* d) {
    return d != nullptr && d->T >= 1 && d->B >= 1 && d->N >= 1 && d->H >= 1 && d->Fin >= 1 && d->M >= 1;
}
size_t eeg_dcrnn_layer_fwd_ws_floats(const eeg_layer_dims* d) {
    if (!layer_dims_positive(d)) return 0;
    const size_t xw = (size_t)d->T * d->B * d->N * 3 * d->H;
    return d->spectral != nullptr ? xw + (size_t)d->N * spec_rows(d->T * d->B) * 3 * d->H : xw;   // + Yh (N, Sp, 3H)
}

size_t eeg_dcrnn_spectral_basis_floats(int N) { return (N >= 1 && N <= kMaxNodes) ? (size_t)spec_basis_floats(N) : 0; }
size_t eeg_dcrnn_spectral_rows(size_t S) { return (S + 15) / 16 * 16; }
int eeg_dcrnn_spectral_basis(const float* support, int N, float* basis, void* stream) {
    if (N < 2 || N > kMaxNodes) return fail("spectral_basis: num_nodes=%d unsupported (2..%d)", N, kMaxNodes);
    if (support == nullptr || basis == nullptr) return fail("spectral_basis: null pointer");
    if (launch_spec_basis(support, N, basis, S_(stream))) return fail("spectral_basis: launch failed");
    return check_launch("spectral_basis");
}
size_t eeg_dcrnn_spectral_pack_floats(int Fin, int H, int M, int N) {
    return (Fin >= 4 && Fin % 4 == 0 && H == 64 && M >= 1 && M <= kMaxM && N >= 1 && N <= kMaxNodes) ? spec_pack_floats(Fin, H, M, N) : 0;
}
int eeg_dcrnn_pack_cell_spectral(const float* Wg, const float* Wc, const float* basis, int Fin, int H, int M, int N, float* spack,
                                 void* stream) {
    if (eeg_dcrnn_spectral_pack_floats(Fin, H, M, N) == 0)
        return fail("pack_cell_spectral: input_dim=%d rnn_units=%d hop matrices=%d nodes=%d: the spectral form exists for 64 units", Fin, H, M, N);
    if (Wg == nullptr || Wc == nullptr || basis == nullptr || spack == nullptr) return fail("pack_cell_spectral: null pointer");
    if (launch_spec_pack(Wg, Wc, basis, Fin, H, M, N, spack, S_(stream))) return fail("pack_cell_spectral: launch failed");
    return check_launch("pack_cell_spectral");
}
int eeg_dcrnn_pack_cells(int n_cells, const float* const* Wg, const float* const* bg, const float* const* Wc, const float* const* bc,
                         const int32_t* Fin, int H, int M, float* const* packs, const float* basis, int N, float* const* spacks,
                         void* stream) {
    if (n_cells < 1 || n_cells > 4) return fail("pack_cells: %d cells (1..4 per launch)", n_cells);
    if (Wg == nullptr || bg == nullptr || Wc == nullptr || bc == nullptr || Fin == nullptr || packs == nullptr) return fail("pack_cells: null pointer table");
    if (!h_supported(H)) return fail("pack_cells: rnn_units=%d unsupported", H);
    if (!m_supported(M)) return fail("pack_cells: num hop matrices M=%d unsupported (1,2,3,4,5,7)", M);
    if ((basis == nullptr) != (spacks == nullptr)) return fail("pack_cells: the per-frequency packs need the basis block and their output table");
    int fin[4];
    for (int c = 0; c < n_cells; ++c) {
        fin[c] = Fin[c];
        if (fin[c] < 4 || fin[c] % 4 != 0) return fail("pack_cells: input dim %d of cell %d must be a positive multiple of 4", fin[c], c);
        if (Wg[c] == nullptr || bg[c] == nullptr || Wc[c] == nullptr || bc[c] == nullptr || packs[c] == nullptr || (spacks != nullptr && spacks[c] == nullptr))
            return fail("pack_cells: null pointer in the tables of cell %d", c);
        if (spacks != nullptr && eeg_dcrnn_spectral_pack_floats(fin[c], H, M, N) == 0)
            return fail("pack_cells: input_dim=%d rnn_units=%d hop matrices=%d nodes=%d: the spectral form exists for 64 units", fin[c], H, M, N);
    }
    if (launch_pack_cells(n_cells, Wg, bg, Wc, bc, fin, H, M, packs, basis, N, spacks, S_(stream))) return fail("pack_cells: launch failed");
    return check_launch("pack_cells");
}
int eeg_dcrnn_spectral_ok(const eeg_layer_dims* d, int need_dx) {
    if (!layer_dims_positive(d) || d->p_batched) return 0;
    if (d->x_planes_ready && d->Fin != d->H) return 0;           // (a handed-over transformed input is the layer below's U^T h)
    if (check_dims(d->N, d->H, d->Fin, d->M)) return 0;
    return spec_supported(d->T, d->B, d->N, d->H, d->Fin, d->M, need_dx) ? 1 : 0;
}

int eeg_dcrnn_batch_major_ok(const eeg_layer_dims* d) {
    if (!diffuse_streams(d->p_batched, d->T * d->B, d->B, d->N, d->Fin)) return 0;
    // 2: the GEMMs read the batch-major input through a row map (d->x_batch_major = 1, no copy); 1: only with the
    // time-major copy Xtm
    const int R = d->T * d->B * d->N;
    return (nn_dma_ok(d->Fin, R, 3 * d->H) && tn_dma_ok(d->Fin, 3 * d->H)) ? 2 : 1;
}

int eeg_dcrnn_layer_fwd(const eeg_layer_dims* d, const float* X, float* Xtm, const float* h0, const float* P,
                        const float* pack, float* planes, float* Hext, float* Rs, float* Us, float* Cs, float* RHs,
                        float* Hplanes, float* RHplanes, float* ws, void* stream) {
    if (check_dims(d->N, d->H, d->Fin, d->M)) return 1;
    if (d->T < 1 || d->B < 1) return fail("layer_fwd: empty sequence/batch (T=%d, B=%d)", d->T, d->B);
    const bool save = Rs != nullptr;
    if (save != (Us != nullptr) || save != (Cs != nullptr) || save != (RHs != nullptr))
        return fail("layer_fwd: Rs/Us/Cs/RHs must be all NULL or all non-NULL");
    hipStream_t st = S_(stream);
    const int S = d->T * d->B, R = S * d->N, H = d->H, M = d->M, Fin = d->Fin;
    const size_t state = (size_t)d->B * d->N * H;
    CellPack p = make_cell_pack(Fin, H, M);
    // slot 0 of Hext = initial state (h0 == NULL: the recurrent kernel clears it)
    if (h0 != nullptr && h0 != Hext && copy_floats(Hext, h0, state, st)) return 1;
    // 1. hoisted diffusion of the layer input: planes[m-1] = P_m X  (Xtm != NULL: X is batch-major and Xtm
    //    receives the time-major copy that everything after this point reads) -- unless the previous layer's
    //    recurrent kernel already left these planes behind (x_planes_ready)
    const size_t xs = d->x_plane_stride > 0 ? (size_t)d->x_plane_stride : (size_t)R * Fin;
    BtMap bt;
    if (d->x_batch_major) {
        if (Xtm != nullptr || d->x_planes_ready) return fail("layer_fwd: x_batch_major excludes Xtm and x_planes_ready");
        if (d->spectral == nullptr && eeg_dcrnn_batch_major_ok(d) != 2) return fail("layer_fwd: x_batch_major is not available for this shape (eeg_dcrnn_batch_major_ok)");
        bt.T = d->T; bt.B = d->B; bt.N = d->N;
    }
    float* XW = ws;
    int rc3 = -1;
    if (d->spectral != nullptr) {
        // spectral form (shared symmetric support; spec_common.h): Xh = U^T X (node-major, time-major rows; kept in `planes` for the
        // backward, or handed over by the layer below: x_planes_ready), Yh_i = Xh_i Wt_i + csum_i * bias (grouped GEMM, K = Fin), and
        // the two-wave recurrent kernel takes Yh as it is (U Yh in its role B) and leaves U^T h / U^T (r*h) behind (Hplanes = Hh
        // (N, B + Sp, H), RHplanes = RHh (N, Sp, H)).  Where that kernel does not apply: XW = U Yh and the by-products as HBM passes.
        if (!eeg_dcrnn_spectral_ok(d, 0)) return fail("layer_fwd: the spectral form does not cover this shape (eeg_dcrnn_spectral_ok)");
        if (Xtm != nullptr || d->spack == nullptr) return fail("layer_fwd: spectral excludes Xtm and needs spack");
        if ((Hplanes != nullptr) != (RHplanes != nullptr)) return fail("layer_fwd: Hplanes/RHplanes must be both NULL or both non-NULL");
        const SpecPack sp = make_spec_pack(Fin, H, M, d->N);
        const int N = d->N, Sp = spec_rows(S), SpE = d->B + Sp;
        const size_t xgs = d->x_plane_stride > 0 ? (size_t)d->x_plane_stride : (size_t)Sp * Fin;
        float* Yh = ws + (size_t)R * 3 * H;
        if (!d->x_planes_ready && launch_spec_mix(1, X, d->spectral, nullptr, N, d->T, d->B, Fin, d->x_batch_major ? 1 : 0, planes, st, "spec_mix_x"))
            return fail("spec_mix: launch failed");
        int nnf = g_tune[20] == 0 ? launch_nnf(planes, xgs, Fin, Sp, N, d->spack + sp.sxr, sp.sxr_stride, Yh, num_cus(), st, "gemm_nn_xw", pack + p.bias,
                                               d->spectral + spec_csum_offset(N)) : -1;
        if (nnf > 0) return fail("gemm_nnf: launch failed");
        if (nnf < 0 && launch_nng(planes, Fin, Sp, N, d->spack + sp.sxq, sp.sxq_stride, sp.nct_x, Yh, num_cus(), st, "gemm_nn_xw", pack + p.bias,
                                  d->spectral + spec_csum_offset(N), xgs)) return fail("gemm_nng: launch failed");
        int done = 0;
        SeqFwdArgs a{XW, h0 != nullptr ? Hext : nullptr, P, d->p_batched, pack + p.bhg, pack + p.bhc, Hext + state, Rs, Us, Cs, RHs, nullptr, nullptr,
                     (size_t)0, d->T, d->B, N, d->act, seq_probe_arg(st)};
        a.variant = g_tune[12] == 0 ? 1 : 0;
        // the two-wave kernel in its SPEC instantiation where it applies (dev knob 22 = 1: the mixes as separate passes)
        if (g_tune[22] == 0 && seq2_spec_ok(H, M, N, d->T, d->B, a.variant, Sp, SpE)) {
            a.spec_U = d->spectral; a.Yh = Yh; a.Hh = Hplanes; a.RHh = RHplanes; a.spec_Sp = Sp; a.spec_SpE = SpE; a.spec_done = &done;
            if (seq_fwd(H, M, a, st)) return 1;
            if (!done) return fail("layer_fwd: the fused spectral recurrent kernel was not selected (internal)");
        }
        if (!done) {
            if (launch_spec_mix(0, Yh, d->spectral, nullptr, N, d->T, d->B, 3 * H, 0, XW, st, "spec_mix_y")) return fail("spec_mix: launch failed");
            if (seq_fwd(H, M, a, st)) return 1;
            if (Hplanes != nullptr) {
                if (launch_spec_mix(1, Hext, d->spectral, nullptr, N, d->T + 1, d->B, H, 0, Hplanes, st, "spec_mix_h", SpE)) return fail("spec_mix: launch failed");
                if (launch_spec_mix(1, RHs, d->spectral, nullptr, N, d->T, d->B, H, 0, RHplanes, st, "spec_mix_h")) return fail("spec_mix: launch failed");
            }
        } else if (Hplanes != nullptr) {
            if (launch_spec_zero_rows(Hplanes, N, (d->T + 1) * d->B, SpE, H, st)) return fail("spec_zero_rows: launch failed");
            if (launch_spec_zero_pad(RHplanes, N, S, H, st)) return fail("spec_zero_pad: launch failed");
        }
        return check_launch("layer_fwd (spectral)");
    } else if (d->x_planes_ready) {
        if (Xtm != nullptr) return fail("layer_fwd: x_planes_ready excludes a batch-major input");
    } else {
        if (diffuse_fwd(X, P, d->p_batched, S, d->B, d->N, Fin, M, planes, st, xs, d->x_batch_major ? 2 : (Xtm != nullptr ? 1 : 0), Xtm)) return 1;
        if (Xtm != nullptr) X = Xtm;
    }
    // 2. hoisted x-part GEMM: XW = [X | planes] @ Bx + [bg|bc]
    SegPtrs segs;
    for (int m = 0; m < kMaxM; ++m) segs.p[m] = m == 0 ? X : (m < M ? planes + (size_t)(m - 1) * xs : nullptr);
    if (rc3 < 0 && d->pack3 != nullptr && pack3_supported(Fin, H, M)) {          // opt-in: three-term bf16 split (include/eeg_dcrnn.h, eeg_layer_dims.pack3)
        const Pack3 q = make_pack3(Fin, H, M);
        rc3 = gemm_nn_bf3(segs, M, Fin, R, d->pack3 + q.xw, q.xw_nct, pack + p.bias, XW, 3 * H, 3 * H, st, "gemm_nn_xw", bt);
        if (rc3 > 0) return 1;
    }
    if (rc3 < 0 && gemm_nn(segs, M, Fin, R, pack + p.bx, 3 * H / 16, pack + p.bias, XW, 3 * H, 3 * H, st, "gemm_nn_xw", bt, p.has_bxq ? pack + p.bxq : nullptr)) return 1;
    // 3. the recurrence
    if ((Hplanes != nullptr) != (RHplanes != nullptr)) return fail("layer_fwd: Hplanes/RHplanes must be both NULL or both non-NULL");
    SeqFwdArgs a{XW, h0 != nullptr ? Hext : nullptr, P, d->p_batched, pack + p.bhg, pack + p.bhc, Hext + state, Rs, Us, Cs, RHs, Hplanes, RHplanes,
                 (size_t)(d->T + 1) * state, d->T, d->B, d->N, d->act, seq_probe_arg(st)};
    a.variant = g_tune[12] == 0 ? 1 : 0;          // two waves per SIMD where that kernel exists (knob 12 = 1: off)
    return seq_fwd(H, M, a, st);
}

size_t eeg_dcrnn_layer_bwd_ws_floats(const eeg_layer_dims* d, int need_dx) { return layer_dims_positive(d) ? bwd_ws(d, need_dx).total : 0; }

int eeg_dcrnn_layer_bwd(const eeg_layer_dims* d, const float* X, const float* P, const float* pack,
                        const float* planes, const float* Hext, const float* Rs, const float* Us, const float* Cs,
                        const float* RHs, const float* Hplanes, const float* RHplanes, const float* dHseq,
                        const float* d_at_end, const float* d_at_len,
                        const int64_t* lengths, float* dX, float* dh0, float* dWg, float* dbg, float* dWc,
                        float* dbc, float* ws, void* stream) {
    if (check_dims(d->N, d->H, d->Fin, d->M)) return 1;
    if (d->T < 1 || d->B < 1) return fail("layer_bwd: empty sequence/batch (T=%d, B=%d)", d->T, d->B);
    hipStream_t st = S_(stream);
    const int S = d->T * d->B, R = S * d->N, H = d->H, M = d->M, Fin = d->Fin, N = d->N;
    const size_t state = (size_t)d->B * N * H;
    CellPack p = make_cell_pack(Fin, H, M);
    BwdWs w = bwd_ws(d, dX != nullptr);
    float* dXW = ws + w.dxw;
    float* dbias = ws + w.dbias;
    // 1. BPTT through the recurrence: dXW = [dR|dU|dC] per step, dh0, per-clip bias partials
    SeqBwdArgs a{Hext + state, Hext, Rs, Us, Cs, dHseq, d_at_end, d_at_len,
                 reinterpret_cast<const long long*>(lengths), P, d->p_batched, pack + p.b1, pack + p.b2,
                 dXW, dh0, dbias, d->T, d->B, N, d->act, seq_probe_arg(st)};
    a.variant = g_tune[13] == 0 ? 1 : 0;          // two waves per SIMD where that kernel exists (knob 13 = 1: off)
    int dyh_done = 0;                             // spectral form: the two-wave BPTT kernel writes dYh = U^T dXW itself (dev knob 21 = 1: separate pass)
    if (d->spectral != nullptr && g_tune[21] == 0 && seq2_spec_ok(H, M, N, d->T, d->B, a.variant, spec_rows(S), d->B + spec_rows(S))) {
        a.spec_U = d->spectral; a.dYh = ws + w.dyh; a.spec_Sp = spec_rows(S); a.spec_done = &dyh_done;
    }
    if (seq_bwd(H, M, a, st)) return 1;
    // 2. weight gradients (hoisted, split-K with fixed-order reduction); the bias sums ride in their reduction launch
    const size_t xs = d->x_plane_stride > 0 ? (size_t)d->x_plane_stride : (size_t)R * Fin;
    BtMap bt;
    if (d->x_batch_major) {
        if (d->spectral == nullptr && eeg_dcrnn_batch_major_ok(d) != 2) return fail("layer_bwd: x_batch_major is not available for this shape");
        bt.T = d->T; bt.B = d->B; bt.N = N;
    }
    if (d->spectral != nullptr) {
        // spectral form: dYh = U^T dXW (node-major, time-major rows; written by the two-wave BPTT kernel itself where it runs),
        // dWt_i = Xh_i^T dYh_i etc. folded into dW_m by the reduction, dX = U [dYh_i Wt_i^T]_i
        if (!eeg_dcrnn_spectral_ok(d, dX != nullptr)) return fail("layer_bwd: the spectral form does not cover this shape (eeg_dcrnn_spectral_ok)");
        if (d->spack == nullptr) return fail("layer_bwd: spectral needs spack");
        if ((Hplanes != nullptr) != (RHplanes != nullptr)) return fail("layer_bwd: Hplanes/RHplanes must be both NULL or both non-NULL");
        const SpecPack sp = make_spec_pack(Fin, H, M, N);
        const int Sp = spec_rows(S), SpE = d->B + Sp;
        float* dYh = ws + w.dyh;
        if (dyh_done) {
            if (launch_spec_zero_pad(dYh, N, S, 3 * H, st)) return fail("spec_zero_pad: launch failed");
        } else if (launch_spec_mix(1, dXW, d->spectral, nullptr, N, d->T, d->B, 3 * H, 0, dYh, st, "spec_mix_dy")) return fail("spec_mix: launch failed");
        const float *hh = Hplanes, *rhh = RHplanes;
        size_t hh_gs = (size_t)SpE * H;
        if (hh == nullptr) {                        // no by-products from the forward: U^T h_{t-1}, U^T (r*h_{t-1}) here
            if (launch_spec_mix(1, Hext, d->spectral, nullptr, N, d->T, d->B, H, 0, ws + w.hplanes, st, "spec_mix_h")) return fail("spec_mix: launch failed");
            if (launch_spec_mix(1, RHs, d->spectral, nullptr, N, d->T, d->B, H, 0, ws + w.rhplanes, st, "spec_mix_h")) return fail("spec_mix: launch failed");
            hh = ws + w.hplanes; rhh = ws + w.rhplanes; hh_gs = 0;
        }
        if (cell_weight_grads_spectral(d, planes, xs_spec(d), hh, hh_gs, rhh, dYh, ws + w.partial, w, dWg, dWc, st, dbias, dbg, dbc)) return 1;
        if (dX != nullptr) {
            // one kernel (GEMM over K = 3H + the node mix back) where it applies; dev knob 17 = 1: the grouped GEMM and the mix as passes
            const int dxf = g_tune[17] == 0 ? launch_dxf(dYh, Sp, S, N, Fin, d->spack + sp.sxtq, sp.sxtq_stride, d->spectral, dX, st, "gemm_dx_f") : -1;
            if (dxf > 0) return fail("gemm_dxf: launch failed");
            if (dxf < 0) {
                float* dXh = ws + w.z;
                if (launch_nng(dYh, 3 * H, Sp, N, d->spack + sp.sxtq, sp.sxtq_stride, sp.nct_t, dXh, num_cus(), st, "gemm_nn_dx")) return fail("gemm_nng: launch failed");
                if (launch_spec_mix(0, dXh, d->spectral, nullptr, N, d->T, d->B, Fin, 0, dX, st, "spec_mix_dx")) return fail("spec_mix: launch failed");
            }
        }
        return check_launch("layer_bwd (spectral)");
    }
    if (cell_weight_grads(d, X, planes, xs, Hext, RHs, dXW, P, Hplanes, RHplanes, (size_t)(d->T + 1) * state,
                          ws + w.hplanes, ws + w.rhplanes, ws + w.partial, w, false, dWg, dWc, st, bt, dbias, dbg, dbc)) return 1;
    // 3. gradient w.r.t. the layer input: Z = dXW @ Bx^T, dX = Z_0 + sum_m P_m^T Z_m
    if (dX != nullptr) {
        float* Z = ws + w.z;
        SegPtrs sd;
        for (int m = 0; m < kMaxM; ++m) sd.p[m] = m == 0 ? dXW : nullptr;
        int rc3 = -1;
        if (d->pack3 != nullptr && pack3_supported(Fin, H, M)) {      // opt-in: three-term bf16 split
            const Pack3 q = make_pack3(Fin, H, M);
            rc3 = gemm_nn_bf3(sd, 1, 3 * H, R, d->pack3 + q.dx, q.dx_nct, nullptr, Z, M * Fin, M * Fin, st, "gemm_nn_dx", BtMap());
            if (rc3 > 0) return 1;
        }
        if (rc3 < 0 && gemm_nn(sd, 1, 3 * H, R, pack + p.bxt, round_up(M * Fin, 16) / 16, nullptr, Z, M * Fin, M * Fin, st, "gemm_nn_dx", BtMap(), p.has_bxtq ? pack + p.bxtq : nullptr)) return 1;
        if (diffuse_adj(Z, P, d->p_batched, S, d->B, N, Fin, M, dX, st)) return 1;
    }
    return 0;
}


/* ---- input featurisation -------------------------------------------------------------------------- */
int eeg_dcrnn_fft_features(const float* raw, int B, int N, int T, int W, const int32_t* perm, const float* log_scale,
                           float mean, float std_, float* feat_raw, float* feat_std, void* stream) {
    if (B < 1 || N < 1 || T < 1) return fail("fft_features: empty input (B=%d, N=%d, T=%d)", B, N, T);
    if (W < 4 || W % 4 != 0 || W / 4 + 1 > 64) return fail("fft_features: window=%d unsupported (multiple of 4, <= 252)", W);
    if (feat_raw == nullptr && feat_std == nullptr) return fail("fft_features: no output requested");
    if (feat_std != nullptr && !(std_ != 0.f)) return fail("fft_features: std must be non-zero");
    if (W == kFftWin) {                               // the reference's 200-sample steps: mixed-radix transform, 6 windows per wave
        const long long n_windows = (long long)B * N * T;
        const long long n_items = (n_windows + kFftPerWave - 1) / kFftPerWave;
        long long blocks = (n_items + 3) / 4;
        const long long cap = (long long)platform_num_cus() * kFftWgPerCu;  // workgroups of 4 waves per CU the kernel's registers allow, persistent
        if (blocks > cap) blocks = cap;
        const size_t lds = 4 * (size_t)kFftWaveDoubles * sizeof(double);
        EEG_SET_MAX_LDS(fft200_features_kernel, lds);
        EEG_LAUNCH_P("fft_features", fft200_features_kernel, dim3((unsigned)blocks), dim3(256), lds, S_(stream), raw, N, T, n_windows,
                     reinterpret_cast<const int*>(perm), log_scale, mean, 1.0f / std_, feat_raw, feat_std);
        return check_launch("fft_features");
    }
    int tchunk = T;                                   // ~8 waves per SIMD in total
    while (tchunk > 1 && (long long)B * N * ceil_div(T, tchunk) < 8192) tchunk = ceil_div(tchunk, 2);
    EEG_LAUNCH_P("fft_features", fft_features_kernel, dim3(B * N, ceil_div(T, tchunk)), dim3(64), (size_t)W * sizeof(double), S_(stream),
                 raw, N, T, W, tchunk, reinterpret_cast<const int*>(perm), log_scale, mean, 1.0f / std_, feat_raw, feat_std);
    return check_launch("fft_features");
}

/* ---- per-clip correlation graph -> supports --------------------------------------------------- */
static int corr_nsplit(int B, int T) {
    int ns = ceil_div(g_tune[7] > 0 ? g_tune[7] : 1024, B);         // dev knob 7: target number of workgroups
    const int cap = ceil_div(T, 4);
    if (ns > cap) ns = cap;
    return ns < 1 ? 1 : ns;
}
int eeg_dcrnn_augment_draw(const uint64_t* rng_used, int B, int N, const int32_t* swap_perm, int32_t* flags, int32_t* perm,
                           float* log_scale, const float* S_plain, const float* S_reflected, int n_supports, float* S_out, void* stream) {
    if (rng_used == nullptr || swap_perm == nullptr || flags == nullptr || perm == nullptr || log_scale == nullptr)
        return fail("augment_draw: null generator pair / swap table / output");
    if (B < 1 || N < 1 || N > kMaxNodes) return fail("augment_draw: B=%d clips, num_nodes=%d unsupported (B >= 1, 1 <= N <= %d)", B, N, kMaxNodes);
    if (S_out != nullptr && (S_plain == nullptr || S_reflected == nullptr || n_supports < 1))
        return fail("augment_draw: per-clip supports need the plain and the reflected set (n_supports=%d)", n_supports);
    EEG_LAUNCH_P("augment_draw", augment_draw_kernel, dim3(B), dim3(128), 0, S_(stream), reinterpret_cast<const unsigned long long*>(rng_used), N,
                 reinterpret_cast<const int*>(swap_perm), reinterpret_cast<int*>(flags), reinterpret_cast<int*>(perm), log_scale, S_plain,
                 S_reflected, n_supports, S_out);
    return check_launch("augment_draw");
}

size_t eeg_dcrnn_corr_graph_ws_floats(int B, int T) { return (B >= 1 && T >= 1) ? (size_t)B * corr_nsplit(B, T) * kGramFloats : 0; }
int eeg_dcrnn_corr_graph(const float* X, int B, int T, int N, int D, int top_k, float* adj, float* S1, float* S2,
                         float* ws, void* stream) {
    if (N < 1 || N > kMaxNodes) return fail("corr_graph: num_nodes=%d unsupported (1..%d)", N, kMaxNodes);
    if (D < 4 || D % 4 != 0) return fail("corr_graph: feature dim=%d unsupported (positive multiple of 4)", D);
    if (B < 1 || T < 1) return fail("corr_graph: empty batch/clip (B=%d, T=%d)", B, T);
    if (top_k < 0 || top_k >= N) return fail("corr_graph: top_k=%d unsupported (0..%d)", top_k, N - 1);
    hipStream_t st = S_(stream);
    const int ns = corr_nsplit(B, T);
    const int step_floats = round_up(N * D, 256);          // one time step, in whole 1-KB wave-DMAs
    const size_t lds = 4 * (size_t)(step_floats > kGramFloats ? step_floats : kGramFloats) * sizeof(float);
    if (lds > 64 * 1024) return fail("corr_graph: N*D=%d too large for the per-wave staging buffers", N * D);
    switch (ceil_div(D, 16)) {
#define EEG_GRAM(NQ)                                                                                                          \
    case NQ:                                                                                                                  \
        if (N <= 20) {                                                                                                        \
            EEG_SET_MAX_LDS((corr_gram_kernel<NQ, true>), lds);                                                               \
            EEG_LAUNCH_P("corr_gram", (corr_gram_kernel<NQ, true>), dim3(B, ns), dim3(256), lds, st, X, T, N, D, ws, step_floats);  \
        } else {                                                                                                              \
            EEG_SET_MAX_LDS((corr_gram_kernel<NQ, false>), lds);                                                              \
            EEG_LAUNCH_P("corr_gram", (corr_gram_kernel<NQ, false>), dim3(B, ns), dim3(256), lds, st, X, T, N, D, ws, step_floats); \
        }                                                                                                                     \
        break;
        EEG_GRAM(1) EEG_GRAM(2) EEG_GRAM(3) EEG_GRAM(4) EEG_GRAM(5) EEG_GRAM(6) EEG_GRAM(7) EEG_GRAM(8)
#undef EEG_GRAM
        default: return fail("corr_graph: feature dim=%d unsupported (<= 128)", D);
    }
    if (check_launch("corr_gram")) return 1;
    EEG_LAUNCH_P("corr_finish", corr_finish_kernel, dim3(B), dim3(256), (2 * 32 * 33 + 64) * sizeof(float), st, ws, ns, N, top_k, adj, S1, S2);
    return check_launch("corr_finish");
}

/* ---- decoder ---------------------------------------------------------------------------------- */
static bool dec_dims_positive(const eeg_decoder_dims* d) {
    return d != nullptr && d->T >= 1 && d->B >= 1 && d->N >= 1 && d->H >= 1 && d->Dout >= 1 && d->M >= 1 && d->L >= 1;
}
size_t eeg_dcrnn_decoder_saved_floats(const eeg_decoder_dims* d) { return dec_dims_positive(d) ? dec_layout(d).saved_total : 0; }
size_t eeg_dcrnn_decoder_fwd_ws_floats(const eeg_decoder_dims* d) { return dec_dims_positive(d) ? (size_t)d->B * d->N * 3 * d->H : 0; }
size_t eeg_dcrnn_decoder_bwd_ws_floats(const eeg_decoder_dims* d) { return dec_dims_positive(d) ? dec_layout(d).bwd_total : 0; }

// the persistent decoder kernels (kernels_decoder.h) cover this shape: forward and backward always pair up over the shared `saved`
// layout (64 units, <= 20 nodes, <= 4 layers, horizon <= 64, Dout <= 128 with Dout/4 divisible by 4 or 5 = the weight-group
// sizes of the backward's projection transpose, both LDS budgets)
static bool dec_persistent_ok(const eeg_decoder_dims* d) {
    const int q4 = d->Dout / 4;
    return d->H == 64 && d->N <= kDecRows && d->L >= 1 && d->L <= 4 && d->T >= 1 && d->T <= 64 && d->Dout <= 128
           && (q4 % 5 == 0 || q4 % 4 == 0) && m_supported(d->M)
           && dec_fwd_lds_floats(d->M, d->L, d->Dout) * sizeof(float) <= kMaxLdsBytes
           && dec_bwd_lds_floats(d->M, d->L, d->Dout) * sizeof(float) <= kMaxLdsBytes;
}
int eeg_dcrnn_decoder_is_persistent(const eeg_decoder_dims* d) {
    if (check_decoder_dims(d)) return 0;
    return dec_persistent_ok(d) ? 1 : 0;
}

int eeg_dcrnn_decoder_fwd(const eeg_decoder_dims* d, const float* targets, const int32_t* teacher, const float* h0,
                          const float* P, const float* const* packs, const float* Wp, const float* bp,
                          const uint64_t* rng_used, float* out, float* saved, float* ws, void* stream) {
    if (check_decoder_dims(d)) return 1;
    ProfPrefix tag("dec_");
    hipStream_t st = S_(stream);
    const DecLayout y = dec_layout(d);
    const DropCfg drop = make_drop_cfg(d->dropout_p);
    const unsigned long long* used = reinterpret_cast<const unsigned long long*>(rng_used);
    if (drop.on && rng_used == nullptr) return fail("decoder_fwd: dropout_p > 0 needs the {seed, offset} pair of eeg_dcrnn_rng_take");
    const bool tf_dev = d->teacher_on_device != 0 && teacher != nullptr;
    if (tf_dev && targets == nullptr) return fail("decoder_fwd: teacher flags on the device need the target sequence");
    const int B = d->B, N = d->N, H = d->H, M = d->M, Dout = d->Dout, L = d->L, RB = B * N;
    const size_t state = (size_t)RB * H, xstep = (size_t)RB * Dout;
    const int nct_o = ceil_div(Dout, 16);
    float* xin = saved + y.xin;
    float* ppack = saved + y.proj_pack;
    float* pbias = saved + y.proj_bias;
    EEG_LAUNCH_P("pack_cell", pack_linear_kernel, dim3(64), dim3(256), 0, st, Wp, Dout, H, 1, ppack);
    if (check_launch("pack_linear")) return 1;
    if (hipMemsetAsync(pbias, 0, (size_t)nct_o * 16 * sizeof(float), st) != hipSuccess) return fail("decoder_fwd: memset failed");
    if (copy_floats(pbias, bp, Dout, st)) return 1;
    if (hipMemsetAsync(xin, 0, xstep * sizeof(float), st) != hipSuccess) return fail("decoder_fwd: memset failed");   // GO symbol
    for (int l = 0; l < L; ++l)
        if (copy_floats(saved + y.hext[l], h0 + (size_t)l * state, state, st)) return 1;
    // ---- persistent path: ONE launch for all T steps, layers and the projection (kernels_decoder.h); where it does not
    //      apply (hidden size, montage > 20 nodes, LDS) the per-step launches below run
    {
        const size_t lds = dec_fwd_lds_floats(M, L, Dout) * sizeof(float);
        if (g_tune[11] == 0 && dec_persistent_ok(d)) {
            DecFwdArgs a;
            for (int l = 0; l < L; ++l) {
                const CellPack p = make_cell_pack(l == 0 ? Dout : H, H, M);
                a.l[l] = DecLayerPtrs{packs[l] + p.bxq, packs[l] + p.bias, packs[l] + p.bhg, packs[l] + p.bhc,
                                      saved + y.hext[l], saved + y.rs[l], saved + y.us[l], saved + y.cs[l], saved + y.rhs[l],
                                      saved + y.hpl[l], saved + y.rpl[l]};
            }
            for (int l = L; l < 4; ++l) a.l[l] = a.l[0];
            a.P = P; a.targets = targets; a.ppack = ppack; a.pbias = pbias;
            a.out = out; a.xin = xin; a.planes0 = saved + y.planes[0];
            a.planes0_stride = (size_t)d->T * RB * Dout;
            a.hplane_stride = (size_t)(d->T + 1) * state;
            a.teacher_mask = 0;
            a.teacher_dev = tf_dev ? reinterpret_cast<const int*>(teacher) : nullptr;
            for (int t = 0; t < d->T && !tf_dev; ++t)
                if (teacher != nullptr && teacher[t] != 0) a.teacher_mask |= 1ull << t;
            a.p_batched = d->p_batched; a.T = d->T; a.B = B; a.N = N; a.Dout = Dout; a.L = L; a.act = d->act;
            a.drop = drop; a.rng_used = used; a.hd = saved + y.hd; a.probe = seq_probe_arg(st);
            const int rc = launch_dec_fwd_persist(M, a, lds, st);
            if (rc == 0) return 0;
            if (rc == 2) return fail("decoder_fwd: persistent kernel launch failed");
        }
    }
    if (tf_dev) return fail("decoder_fwd: teacher_on_device needs the persistent decoder kernel, which does not cover this shape "
                            "(H=%d, N=%d, L=%d, T=%d, Dout=%d, M=%d): pass the flags as a host array", H, N, L, d->T, Dout, M);
    float* XW = ws;
    for (int t = 0; t < d->T; ++t) {
        for (int l = 0; l < L; ++l) {
            const int Fin = l == 0 ? Dout : H;
            const size_t Rall = (size_t)d->T * RB, hstride = (size_t)(d->T + 1) * state;
            const float* X = l == 0 ? xin + (size_t)t * xstep : saved + y.hext[l - 1] + (size_t)(t + 1) * state;
            // input hop planes: layer 0 diffuses its input; above, the layer below has just left them behind
            float* planes = l == 0 ? saved + y.planes[0] + (size_t)t * RB * Fin : saved + y.hpl[l - 1] + (size_t)(t + 1) * state;
            const size_t xs = l == 0 ? Rall * Fin : hstride;
            const CellPack p = make_cell_pack(Fin, H, M);
            const float* pack = packs[l];
            if (l == 0 && diffuse_fwd(X, P, d->p_batched, B, B, N, Fin, M, planes, st, xs)) return 1;
            SegPtrs segs;
            for (int m = 0; m < kMaxM; ++m) segs.p[m] = m == 0 ? X : (m < M ? planes + (size_t)(m - 1) * xs : nullptr);
            if (gemm_nn(segs, M, Fin, RB, pack + p.bx, 3 * H / 16, pack + p.bias, XW, 3 * H, 3 * H, st)) return 1;
            float* Hext = saved + y.hext[l];
            SeqFwdArgs a{XW, Hext + (size_t)t * state, P, d->p_batched, pack + p.bhg, pack + p.bhc,
                         Hext + (size_t)(t + 1) * state, saved + y.rs[l] + (size_t)t * state,
                         saved + y.us[l] + (size_t)t * state, saved + y.cs[l] + (size_t)t * state,
                         saved + y.rhs[l] + (size_t)t * state, saved + y.hpl[l] + (size_t)t * state,
                         saved + y.rpl[l] + (size_t)t * state, hstride, 1, B, N, d->act, nullptr};
            a.variant = g_tune[12] == 0 ? 1 : 0;
            if (seq_fwd(H, M, a, st)) return 1;
        }
        // projection (model.py:188-191): out_t = drop(h_top) W_p^T + b_p
        if (drop.on) {
            EEG_LAUNCH_P("dropout_apply", dropout_apply_kernel, dim3(ceil_div((int)(state / 4), 256)), dim3(256), 0, st,
                         saved + y.hext[L - 1] + (size_t)(t + 1) * state, saved + y.hd + (size_t)t * state, state, (size_t)t * state, used, drop);
            if (check_launch("dropout_apply")) return 1;
        }
        SegPtrs sp;
        for (int m = 0; m < kMaxM; ++m) sp.p[m] = m == 0 ? saved + y.hd + (size_t)t * state : nullptr;
        if (gemm_nn(sp, 1, H, RB, ppack, nct_o, pbias, out + (size_t)t * xstep, Dout, Dout, st)) return 1;
        if (t + 1 < d->T) {   // next decoder input: the projection, or the target under teacher forcing (model.py:194-200)
            const bool tf = teacher != nullptr && teacher[t] != 0;
            if (copy_floats(xin + (size_t)(t + 1) * xstep, tf ? targets + (size_t)t * xstep : out + (size_t)t * xstep, xstep, st)) return 1;
        }
    }
    return 0;
}

int eeg_dcrnn_decoder_bwd(const eeg_decoder_dims* d, const int32_t* teacher, const float* P, const float* const* packs,
                          const float* Wp, const float* saved, const float* dOut, const uint64_t* rng_used, float* dh0,
                          float* const* dWg, float* const* dbg, float* const* dWc, float* const* dbc, float* dWp, float* dbp,
                          float* ws, void* stream) {
    if (check_decoder_dims(d)) return 1;
    ProfPrefix tag("dec_");
    hipStream_t st = S_(stream);
    const DecLayout y = dec_layout(d);
    const DropCfg drop = make_drop_cfg(d->dropout_p);
    const unsigned long long* used = reinterpret_cast<const unsigned long long*>(rng_used);
    if (drop.on && rng_used == nullptr) return fail("decoder_bwd: dropout_p > 0 needs the rng_used pair of the forward call");
    const bool tf_dev = d->teacher_on_device != 0 && teacher != nullptr;
    const int B = d->B, N = d->N, H = d->H, M = d->M, Dout = d->Dout, L = d->L, RB = B * N, T = d->T;
    const size_t state = (size_t)RB * H, xstep = (size_t)RB * Dout, Rall = (size_t)T * RB;
    const int nct_h = ceil_div(H, 16);
    float* tpack = ws + y.projt_pack;
    EEG_LAUNCH_P("pack_cell", pack_linear_kernel, dim3(64), dim3(256), 0, st, Wp, Dout, H, 0, tpack);
    if (check_launch("pack_linear")) return 1;
    float* dOtot = ws + y.dotot;
    float* dA = ws + y.da;
    float* Z = ws + y.z;
    // out_t is step t+1's input (host flags; with device flags the persistent kernel derives the same mask itself)
    auto feeds_back = [&](int t) { return t + 1 < T && !(!tf_dev && teacher != nullptr && teacher[t] != 0); };
    // ---- persistent path: ONE launch walks the T steps backwards (kernels_decoder.h); it leaves dXW of every
    //      (layer, step), dOtot and dh0 -- the hoisted parameter gradients below are common to both paths
    bool persistent = false;
    {
        const int q4 = Dout / 4;
        const int dt = q4 % 5 == 0 ? 5 : (q4 % 4 == 0 ? 4 : 0);
        const size_t lds = dec_bwd_lds_floats(M, L, Dout) * sizeof(float);
        if (g_tune[10] == 0 && dec_persistent_ok(d)) {
            DecBwdArgs a;
            for (int l = 0; l < L; ++l) {
                const CellPack p = make_cell_pack(l == 0 ? Dout : H, H, M);
                a.l[l] = DecBwdLayerPtrs{packs[l] + p.c1, packs[l] + p.c2, saved + y.hext[l], saved + y.rs[l],
                                         saved + y.us[l], saved + y.cs[l], ws + y.dxw[l]};
            }
            for (int l = L; l < 4; ++l) a.l[l] = a.l[0];
            a.P = P; a.tpack = tpack; a.dOut = dOut; a.dOtot = dOtot; a.dh0 = dh0;
            a.dbias0 = ws + y.dbias[0]; a.dbias1 = ws + y.dbias[L > 1 ? 1 : 0];
            a.feeds_mask = 0;
            a.teacher_dev = tf_dev ? reinterpret_cast<const int*>(teacher) : nullptr;
            for (int t = 0; t < T; ++t)
                if (feeds_back(t)) a.feeds_mask |= 1ull << t;
            a.p_batched = d->p_batched; a.T = T; a.B = B; a.N = N; a.Dout = Dout; a.L = L; a.act = d->act;
            a.drop = drop; a.rng_used = used; a.probe = seq_probe_arg(st);
            const int rc = launch_dec_bwd_persist(M, dt, a, lds, st);
            if (rc == 2) return fail("decoder_bwd: persistent kernel launch failed");
            persistent = rc == 0;
        }
    }
    if (tf_dev && !persistent) return fail("decoder_bwd: teacher_on_device needs the persistent decoder kernel, which does not cover "
                                           "this shape (H=%d, N=%d, L=%d, T=%d, Dout=%d, M=%d)", H, N, L, T, Dout, M);
    for (int t = persistent ? -1 : T - 1; t >= 0; --t) {
        // total gradient of out_t: the loss term, plus (autoregressive feedback) dx of layer 0 at step t+1, which
        // the adjoint diffusion of that step has already added into dOtot[t]
        const float* dO = feeds_back(t) ? dOtot + (size_t)t * xstep : dOut + (size_t)t * xstep;
        if (!feeds_back(t) && copy_floats(dOtot + (size_t)t * xstep, dO, xstep, st)) return 1;   // dense copy for the hoisted GEMM
        SegPtrs sp;
        for (int m = 0; m < kMaxM; ++m) sp.p[m] = m == 0 ? dOtot + (size_t)t * xstep : nullptr;
        if (gemm_nn(sp, 1, Dout, RB, tpack, nct_h, nullptr, dA, H, H, st)) return 1;               // d drop(h_top) = dO W_p
        if (drop.on) {                                                                              // d h_top = mask * that
            EEG_LAUNCH_P("dropout_apply", dropout_apply_kernel, dim3(ceil_div((int)(state / 4), 256)), dim3(256), 0, st, dA, dA, state, (size_t)t * state, used, drop);
            if (check_launch("dropout_apply")) return 1;
        }
        for (int l = L - 1; l >= 0; --l) {
            const int Fin = l == 0 ? Dout : H;
            const CellPack p = make_cell_pack(Fin, H, M);
            const float* pack = packs[l];
            const float* Hext = saved + y.hext[l];
            float* dXW = ws + y.dxw[l] + (size_t)t * RB * 3 * H;
            float* dhn_in = ws + y.dhn + ((size_t)((t + 1) & 1) * L + l) * state;
            float* dhn_out = ws + y.dhn + ((size_t)(t & 1) * L + l) * state;
            SeqBwdArgs a{Hext + (size_t)(t + 1) * state, Hext + (size_t)t * state, saved + y.rs[l] + (size_t)t * state,
                         saved + y.us[l] + (size_t)t * state, saved + y.cs[l] + (size_t)t * state, dA,
                         t == T - 1 ? nullptr : dhn_in, nullptr, nullptr, P, d->p_batched, pack + p.b1, pack + p.b2,
                         dXW, t == 0 ? dh0 + (size_t)l * state : dhn_out, ws + y.dbias[l] + (size_t)t * B * 3 * H,
                         1, B, N, d->act, nullptr};
            a.variant = g_tune[13] == 0 ? 1 : 0;
            if (seq_bwd(H, M, a, st)) return 1;
            const bool need_dx = l > 0 || (t > 0 && feeds_back(t - 1));
            if (need_dx) {
                SegPtrs sd;
                for (int m = 0; m < kMaxM; ++m) sd.p[m] = m == 0 ? dXW : nullptr;
                if (gemm_nn(sd, 1, 3 * H, RB, pack + p.bxt, round_up(M * Fin, 16) / 16, nullptr, Z, M * Fin, M * Fin, st)) return 1;
                if (l > 0) {
                    if (diffuse_adj(Z, P, d->p_batched, B, B, N, Fin, M, dA, st)) return 1;         // -> layer below's h_t
                } else {
                    if (diffuse_adj(Z, P, d->p_batched, B, B, N, Fin, M, dOtot + (size_t)(t - 1) * xstep, st,
                                    dOut + (size_t)(t - 1) * xstep)) return 1;
                }
            }
        }
    }
    // hoisted parameter gradients over all T steps
    for (int l = 0; l < L; ++l) {
        const bool first_use = l <= 1;                    // layers >= 1 share one cell (model.py:126-143)
        const float* X = l == 0 ? saved + y.xin : saved + y.hext[l - 1] + state;
        const size_t hstride = (size_t)(T + 1) * state;
        const float* xpl = l == 0 ? saved + y.planes[0] : saved + y.hpl[l - 1] + state;
        if (cell_weight_grads(&y.ld[l], X, xpl, l == 0 ? Rall * Dout : hstride, saved + y.hext[l], saved + y.rhs[l],
                              ws + y.dxw[l], P, saved + y.hpl[l], saved + y.rpl[l], hstride, nullptr, nullptr,
                              ws + y.partial, y.lw[l], !first_use, dWg[l], dWc[l], st)) return 1;
    }
    if (persistent) {   // per-clip sums over steps and nodes (layers >= 1 share a cell: already added up)
        if (colsum(ws + y.dbias[0], B, 3 * H, 2 * H, ws + y.colsum, dbg[0], dbc[0], st)) return 1;
        if (L > 1 && colsum(ws + y.dbias[1], B, 3 * H, 2 * H, ws + y.colsum, dbg[1], dbc[1], st)) return 1;
    } else {
        if (colsum(ws + y.dbias[0], T * B, 3 * H, 2 * H, ws + y.colsum, dbg[0], dbc[0], st)) return 1;
        if (L > 1 && colsum(ws + y.dbias[1], (L - 1) * T * B, 3 * H, 2 * H, ws + y.colsum, dbg[1], dbc[1], st)) return 1;
    }
    // projection: dW_p (Dout x H) = sum_rows dOtot^T h_top ;  db_p = column sums of dOtot
    SegPtrs so;
    for (int m = 0; m < kMaxM; ++m) so.p[m] = m == 0 ? dOtot : nullptr;
    if (gemm_tn(so, 1, Dout, (int)Rall, saved + y.hd, H, 0, H, ws + y.partial, y.nsplit_p, y.rps_p, st)) return 1;   // (hd = h_top without dropout)
    EEG_LAUNCH_P("reduce_unpack", reduce_unpack_kernel, dim3(ceil_div(Dout * H, 64)), dim3(256), 256 * sizeof(float4), st, ws + y.partial, y.nsplit_p, Dout, H, 3, Dout, H, 1, dWp, dWp);
    if (check_launch("reduce_unpack(proj)")) return 1;
    return colsum(dOtot, (int)Rall, Dout, Dout, ws + y.colsum, dbp, nullptr, st);
}

int eeg_dcrnn_gather_last(const float* Htop, const int64_t* lengths, int T, int B, int NH, float* last, void* stream) {
    if (T < 1 || B < 1 || NH < 1) return fail("gather_last: empty input (T=%d, B=%d, N*H=%d)", T, B, NH);
    EEG_LAUNCH_P("gather_last", gather_last_kernel, dim3(ceil_div(B * NH, 256)), dim3(256), 0, S_(stream), Htop,
               reinterpret_cast<const long long*>(lengths), T, B, NH, last);
    return check_launch("gather_last");
}
int eeg_dcrnn_dropout_mask(const uint64_t* rng_used, size_t n, float dropout_p, float* mask, void* stream) {
    if (check_dropout_p("dropout_mask", dropout_p)) return 1;
    const DropCfg drop = make_drop_cfg(dropout_p);
    if (drop.on && rng_used == nullptr) return fail("dropout_mask: dropout_p > 0 needs the rng_used pair of a forward call");
    if (n == 0) return 0;
    EEG_LAUNCH_P("dropout_mask", dropout_mask_kernel, dim3(256), dim3(256), 0, S_(stream), reinterpret_cast<const unsigned long long*>(rng_used), n, drop, mask);
    return check_launch("dropout_mask");
}
int eeg_dcrnn_rng_take(uint64_t* rng_state, uint64_t groups, uint64_t* rng_used, void* stream) {
    if (rng_state == nullptr || rng_used == nullptr) return fail("rng_take: null state / output");
    EEG_LAUNCH_P("rng_take", rng_take_kernel, dim3(1), dim3(64), 0, S_(stream), reinterpret_cast<unsigned long long*>(rng_state),
                 reinterpret_cast<unsigned long long*>(rng_used), (unsigned long long)groups);
    return check_launch("rng_take");
}
int eeg_dcrnn_teacher_flags(uint64_t* rng_state, int64_t* samples_seen, int64_t increment, double cl_decay_steps, int T,
                            int32_t* flags, void* stream) {
    if (rng_state == nullptr || samples_seen == nullptr || flags == nullptr) return fail("teacher_flags: null state / counter / output");
    if (T < 1 || T > 64) return fail("teacher_flags: T=%d unsupported (1..64)", T);
    if (!(cl_decay_steps > 0.0)) return fail("teacher_flags: cl_decay_steps must be positive");
    EEG_LAUNCH_P("teacher_flags", teacher_flags_kernel, dim3(1), dim3(64), 0, S_(stream), reinterpret_cast<unsigned long long*>(rng_state),
                 reinterpret_cast<long long*>(samples_seen), (long long)increment, cl_decay_steps, T, reinterpret_cast<int*>(flags));
    return check_launch("teacher_flags");
}
int eeg_dcrnn_cls_head_fwd(const float* z, const float* W, const float* bias, int B, int N, int H, int C, float dropout_p,
                           const uint64_t* rng_used, float* logits, int32_t* arg, void* stream) {
    if (N > 64) return fail("cls_head: num_nodes=%d unsupported (<= 64)", N);
    if (H % 4 != 0) return fail("cls_head: rnn_units=%d must be a multiple of 4", H);
    if (check_dropout_p("cls_head", dropout_p)) return 1;
    const DropCfg drop = make_drop_cfg(dropout_p);
    if (drop.on && rng_used == nullptr) return fail("cls_head: dropout_p > 0 needs the {seed, offset} pair of eeg_dcrnn_rng_take");
    // LDS: [N][C] node logits + [N][H] masked relu(z) rows
    const size_t head_lds = (size_t)N * (C + H) * sizeof(float);
    if (B < 1 || C < 1 || head_lds > kMaxLdsBytes) return fail("cls_head: B=%d, N=%d x (classes=%d + rnn_units=%d) floats do not fit the %zu-byte LDS", B, N, C, H, (size_t)kMaxLdsBytes);
    EEG_SET_MAX_LDS(cls_head_fwd_kernel, head_lds);
    EEG_LAUNCH_P("cls_head_fwd", cls_head_fwd_kernel, dim3(B), dim3(64), head_lds, S_(stream), z, W, bias, B, N, H, C, drop,
                 reinterpret_cast<const unsigned long long*>(rng_used), logits, arg);
    return check_launch("cls_head_fwd");
}
int eeg_dcrnn_cls_head_bwd(const float* z, const float* W, const float* dlogits, const int32_t* arg, int B, int N,
                           int H, int C, float dropout_p, const uint64_t* rng_used, float* dz, float* dW, float* dbias, void* stream) {
    if (check_dropout_p("cls_head_bwd", dropout_p)) return 1;
    const DropCfg drop = make_drop_cfg(dropout_p);
    if (drop.on && rng_used == nullptr) return fail("cls_head_bwd: dropout_p > 0 needs the rng_used pair of the forward call");
    if (B < 1 || N < 1 || H < 1 || C < 1) return fail("cls_head_bwd: empty input (B=%d, N=%d, H=%d, C=%d)", B, N, H, C);
    const unsigned long long* used = reinterpret_cast<const unsigned long long*>(rng_used);
    EEG_LAUNCH_P("cls_head_bwd_dz", cls_head_bwd_dz_kernel, dim3(ceil_div(B * N * H, 256)), dim3(256), 0, S_(stream), z, W, dlogits, arg, B, N, H, C, drop, used, dz);
    if (check_launch("cls_head_bwd_dz")) return 1;
    EEG_LAUNCH_P("cls_head_bwd_w", cls_head_bwd_w_kernel, dim3(ceil_div(C * H + C, 16)), dim3(256), 256 * sizeof(float), S_(stream), z, dlogits, arg, B, N, H, C, drop, used, dW, dbias);
    return check_launch("cls_head_bwd_w");
}

size_t eeg_dcrnn_cls_head_loss_ws_floats(int B, int H, int C) {
    return (B >= 1 && H >= 1 && C >= 1) ? (size_t)ceil_div(B, 4) * ((size_t)C * H + C + 1) : 0;
}
int eeg_dcrnn_cls_head_loss(const float* z, const float* W, const float* bias, const void* targets, int kind, int B, int N, int H, int C,
                            float dropout_p, const uint64_t* rng_used, float* logits, int32_t* arg, float* dlogits, float* dz,
                            float* dW, float* dbias, float* loss, float* ws, void* stream) {
    if (B < 1 || N < 1 || H < 1 || C < 1) return fail("cls_head_loss: empty input (B=%d, N=%d, H=%d, C=%d)", B, N, H, C);
    if (N > 64) return fail("cls_head_loss: num_nodes=%d unsupported (<= 64)", N);
    if (H % 4 != 0) return fail("cls_head_loss: rnn_units=%d must be a multiple of 4", H);
    if (kind != 0 && kind != 1) return fail("cls_head_loss: kind=%d (0 = BCE-with-logits, 1 = cross-entropy)", kind);
    if (kind == 0 && C != 1) return fail("cls_head_loss: BCE-with-logits takes one logit per clip (num_classes=%d)", C);
    if (check_dropout_p("cls_head_loss", dropout_p)) return 1;
    const DropCfg drop = make_drop_cfg(dropout_p);
    if (drop.on && rng_used == nullptr) return fail("cls_head_loss: dropout_p > 0 needs the {seed, offset} pair of eeg_dcrnn_rng_take");
    if (z == nullptr || W == nullptr || bias == nullptr || targets == nullptr || logits == nullptr || arg == nullptr || dlogits == nullptr ||
        dz == nullptr || dW == nullptr || dbias == nullptr || loss == nullptr || ws == nullptr)
        return fail("cls_head_loss: null pointer");
    const int O = C * H + C, nblk = ceil_div(B, 4);
    const size_t lds = (size_t)(4 * cls_tail_wave_floats(N, H, C) + 4 * (O + 1)) * sizeof(float);
    if (lds > kMaxLdsBytes) return fail("cls_head_loss: N=%d x (classes=%d + rnn_units=%d) floats of four clips do not fit the %zu-byte LDS", N, C, H, (size_t)kMaxLdsBytes);
    EEG_SET_MAX_LDS(cls_head_loss_kernel, lds);
    EEG_LAUNCH_P("cls_head_loss", cls_head_loss_kernel, dim3(nblk), dim3(256), lds, S_(stream), z, W, bias, targets, kind, B, N, H, C, drop,
                 reinterpret_cast<const unsigned long long*>(rng_used), logits, reinterpret_cast<int*>(arg), dlogits, dz, ws);
    if (check_launch("cls_head_loss")) return 1;
    EEG_LAUNCH_P("cls_head_loss", cls_head_loss_finish_kernel, dim3(1), dim3(256), 0, S_(stream), ws, nblk, B, H, C, dW, dbias, loss);
    return check_launch("cls_head_loss_finish");
}
size_t eeg_dcrnn_dconv_fwd_ws_floats(int B, int N, int F, int M, int O) {
    if (B < 1 || N < 1 || F < 1 || M < 1 || O < 1) return 0;
    return (size_t)(M - 1) * B * N * F + (size_t)F * M * O;
}
int eeg_dcrnn_dconv_fwd(const float* X, const float* P, int p_batched, int B, int N, int F, int M,
                        const float* W, const float* bias, int O, float* out, float* ws, void* stream) {
    if (N < 1 || N > kMaxNodes || F < 4 || F % 4 != 0 || M < 1 || M > kMaxM) return fail("dconv_fwd: bad dims N=%d F=%d M=%d", N, F, M);
    if (B < 1) return fail("dconv_fwd: empty batch (B=%d)", B);
    if (O % 16 != 0) return fail("dconv_fwd: output_dim=%d must be a multiple of 16", O);
    hipStream_t st = S_(stream);
    float* planes = ws;
    float* pack = ws + (size_t)(M - 1) * B * N * F;
    if (diffuse_fwd(X, P, p_batched, B, B, N, F, M, planes, st)) return 1;
    EEG_LAUNCH_P("pack_dense", pack_dense_kernel, dim3(256), dim3(256), 0, st, W, F, M, O, pack);
    if (check_launch("pack_dense")) return 1;
    SegPtrs segs;
    for (int m = 0; m < kMaxM; ++m) segs.p[m] = m == 0 ? X : (m < M ? planes + (size_t)(m - 1) * B * N * F : nullptr);
    return gemm_nn(segs, M, F, B * N, pack, O / 16, bias, out, O, O, st);
}
size_t eeg_dcrnn_dconv_bwd_ws_floats(int B, int N, int F, int M, int O) {
    if (B < 1 || N < 1 || F < 1 || M < 1 || O < 1) return 0;
    const int R = B * N;
    int rps;
    const int nsplit = tn_split(M, F, R, O, &rps);
    return (size_t)(M - 1) * R * F + align64((size_t)nsplit * M * F * O) + align64((size_t)(O / 4) * (round_up(M * F, 16) / 16) * 64)
           + (size_t)R * M * F + align64(colsum_ws(R, O));
}
int eeg_dcrnn_dconv_bwd(const float* X, const float* P, int p_batched, int B, int N, int F, int M, const float* W, int O,
                        const float* dOut, float* dX, float* dW, float* dbias, float* ws, void* stream) {
    if (N < 1 || N > kMaxNodes || F < 4 || F % 4 != 0 || M < 1 || M > kMaxM) return fail("dconv_bwd: bad dims N=%d F=%d M=%d", N, F, M);
    if (B < 1) return fail("dconv_bwd: empty batch (B=%d)", B);
    if (O % 16 != 0 || O > 192) return fail("dconv_bwd: output_dim=%d must be a multiple of 16 (<= 192)", O);
    hipStream_t st = S_(stream);
    const int R = B * N;
    int rps;
    const int nsplit = tn_split(M, F, R, O, &rps);
    float* planes = ws;
    float* part = planes + (size_t)(M - 1) * R * F;
    float* tpack = part + align64((size_t)nsplit * M * F * O);
    float* Z = tpack + align64((size_t)(O / 4) * (round_up(M * F, 16) / 16) * 64);
    float* cws = Z + (size_t)R * M * F;
    // dW[f*M+m][o] = sum_rows (P_m X)[row][f] dOut[row][o]: the hop planes are re-formed here (nothing is kept by the forward)
    if (dW != nullptr) {
        if (diffuse_fwd(X, P, p_batched, B, B, N, F, M, planes, st)) return 1;
        SegPtrs sx;
        for (int m = 0; m < kMaxM; ++m) sx.p[m] = m == 0 ? X : (m < M ? planes + (size_t)(m - 1) * R * F : nullptr);
        if (gemm_tn(sx, M, F, R, dOut, O, 0, O, part, nsplit, rps, st)) return 1;
        EEG_LAUNCH_P("reduce_unpack", reduce_unpack_kernel, dim3(ceil_div(M * F * O, 64)), dim3(256), 256 * sizeof(float4), st, part, nsplit, M * F, O, 4, F, 0, M, dW, dW);
        if (check_launch("reduce_unpack(dconv)")) return 1;
    }
    if (dbias != nullptr && colsum(dOut, R, O, O, cws, dbias, nullptr, st)) return 1;
    // dX = Z_0 + sum_m P_m^T Z_m with Z = dOut W^T (R x M*F, hop-major)
    if (dX != nullptr) {
        EEG_LAUNCH_P("pack_dense", pack_dense_t_kernel, dim3(256), dim3(256), 0, st, W, F, M, O, tpack);
        if (check_launch("pack_dense_t")) return 1;
        SegPtrs sd;
        for (int m = 0; m < kMaxM; ++m) sd.p[m] = m == 0 ? dOut : nullptr;
        if (gemm_nn(sd, 1, O, R, tpack, round_up(M * F, 16) / 16, nullptr, Z, M * F, M * F, st)) return 1;
        if (diffuse_adj(Z, P, p_batched, B, B, N, F, M, dX, st)) return 1;
    }
    return 0;
}
int eeg_dcrnn_bce_logits(const float* logits, const float* y, int B, float* loss, float* dlogits, void* stream) {
    if (B < 1) return fail("bce_logits: empty batch");
    EEG_LAUNCH_P("loss_bce", bce_logits_kernel, dim3(1), dim3(256), 256 * sizeof(float), S_(stream), logits, y, B, loss, dlogits);
    return check_launch("bce_logits");
}
int eeg_dcrnn_ce_logits(const float* logits, const int64_t* y, int B, int C, float* loss, float* dlogits, void* stream) {
    if (B < 1 || C < 1) return fail("ce_logits: empty batch");
    EEG_LAUNCH_P("loss_ce", ce_logits_kernel, dim3(1), dim3(256), 256 * sizeof(float), S_(stream), logits,
                 reinterpret_cast<const long long*>(y), B, C, loss, dlogits);
    return check_launch("ce_logits");
}
size_t eeg_dcrnn_masked_loss_ws_floats(void) { return 2 * kLossBlocks + 64; }
int eeg_dcrnn_masked_loss(const float* pred, const float* y, size_t n, int use_scaler, float mean, float std_,
                          float mask_val, int kind, float* loss, float* dpred, float* ws, void* stream) {
    if (n < 1) return fail("masked_loss: empty tensors");
    if (kind != 0 && kind != 1) return fail("masked_loss: kind=%d unsupported (0 = MAE, 1 = RMSE)", kind);
    hipStream_t st = S_(stream);
    int nblk = (int)((n + 256 * 8 - 1) / (256 * 8));
    if (nblk > kLossBlocks) nblk = kLossBlocks;
    // 16-byte accesses need 16-byte aligned tensors (torch allocations are; views with an odd offset are not)
    if (((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dpred)) & 15) != 0)
        return fail("masked_loss: tensors must be 16-byte aligned");
    EEG_LAUNCH_P("loss_masked", masked_loss_partial_kernel, dim3(nblk), dim3(256), 512 * sizeof(float), st, pred, y, n, mean, std_, use_scaler, mask_val, kind, ws);
    if (check_launch("masked_loss_partial")) return 1;
    EEG_LAUNCH_P("loss_masked", masked_loss_finish_kernel, dim3(1), dim3(256), 512 * sizeof(float), st, ws, nblk, kind, loss);
    if (check_launch("masked_loss_finish")) return 1;
    if (dpred != nullptr) {
        EEG_LAUNCH_P("loss_masked", masked_loss_grad_kernel, dim3(nblk * 2), dim3(256), 0, st, pred, y, n, mean, std_, use_scaler, mask_val, kind, ws, dpred);
        if (check_launch("masked_loss_grad")) return 1;
    }
    return 0;
}
size_t eeg_dcrnn_clip_adam_ws_floats(void) { return 64; }
static int clip_adam_launch(float* params, float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float max_norm, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, float* ws,
                            float* norm_out, int32_t* step_dev, const float* lr_dev, void* stream) {
    const int nparts = 64;
    if (n < 1) return fail("clip_adam: empty parameter buffer");
    EEG_LAUNCH_P("grad_sqnorm", sqnorm_partial_kernel, dim3(nparts), dim3(256), 256 * sizeof(float), S_(stream), grads, n, ws, reinterpret_cast<int*>(step_dev));
    if (check_launch("grad_sqnorm")) return 1;
    // torch.optim.Adam forms `1 - beta ** step` and its square root as Python (fp64) scalars
    const float bc1 = step_dev ? 1.f : (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = step_dev ? 1.f : (float)sqrt(1.0 - pow((double)beta2, (double)step));
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    EEG_LAUNCH_P("clip_adam", clip_adam_kernel, dim3(blocks), dim3(256), 0, S_(stream), params, grads, exp_avg, exp_avg_sq, n,
                 ws, nparts, max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, norm_out,
                 reinterpret_cast<const int*>(step_dev), lr_dev);
    return check_launch("clip_adam");
}
int eeg_dcrnn_clip_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float max_norm,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                        float grad_scale, float* ws, float* norm_out, void* stream) {
    if (step < 1) return fail("clip_adam: step must be >= 1");
    return clip_adam_launch(params, grads, exp_avg, exp_avg_sq, n, max_norm, lr, beta1, beta2, eps, weight_decay, step, grad_scale, ws,
                            norm_out, nullptr, nullptr, stream);
}
int eeg_dcrnn_clip_adam_dev(float* params, float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float max_norm,
                            const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, int32_t* step_dev,
                            float grad_scale, float* ws, float* norm_out, void* stream) {
    if (lr_dev == nullptr || step_dev == nullptr) return fail("clip_adam_dev: null learning-rate / step-counter pointer");
    return clip_adam_launch(params, grads, exp_avg, exp_avg_sq, n, max_norm, 0.f, beta1, beta2, eps, weight_decay, 0, grad_scale, ws,
                            norm_out, step_dev, lr_dev, stream);
}

}  // extern "C"
