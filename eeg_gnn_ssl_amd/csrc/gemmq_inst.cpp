// Instantiations + launch logic of the round-3 hoisted GEMMs (kernels_gemm_q.h), in their own translation unit.
#include "kernels_gemm_q.h"
#include "gemmq_launch.h"
#include "prof.h"
#include "seq_launch.h"

namespace eeg {

bool nnq_supported(int nseg, int F, int R, int nct_total, int ldc, int O) {
    if (nseg < 1 || nseg > kMaxM || F < 4 || F % 4 != 0 || R < 1) return false;
    if (nct_total < 12 || nct_total % 12 != 0 || O % 4 != 0 || ldc % 4 != 0 || O > 16 * nct_total) return false;
    if (make_nnq_order(nseg, F).ntail > 2) return false;
    // operands and results go through 2 GB buffer descriptors (platform.h make_wbuf): accesses beyond are dropped by the hardware
    return (double)R * F * 4.0 < 2147483648.0 && (double)R * ldc * 4.0 < 2147483648.0;
}
size_t nnq_pack_floats(int nseg, int F, int nct) { return (size_t)make_nnq_order(nseg, F).nch * nct * 256; }

int launch_nnq(const SegPtrs& segs, int nseg, int F, int R, const float* Bq, int nct_total, const float* bias, float* C,
               int ldc, int O, int btT, int btB, int btN, int num_cus, hipStream_t st, const char* tag) {
    constexpr int NS = 4;                                  // gemm_nnr_kernel: 4 activation stages of 8 KB + 192 bias floats
    const size_t lds = ((size_t)NS * 128 * 16 + 192) * sizeof(float);
    EEG_SET_MAX_LDS((gemm_nnr_kernel<NS, 2>), lds);
    const int rt = ceil_div(R, 16);
    int G = 2 * (num_cus > 0 ? num_cus : 256);
    if (G > ceil_div(rt, 8)) G = ceil_div(rt, 8);          // at least one 128-row tile per workgroup
    if (G < 1) G = 1;
    EEG_LAUNCH_P(tag, (gemm_nnr_kernel<NS, 2>), dim3(G, nct_total / 12), dim3(256), lds, st, segs, nseg, F, R, Bq, nct_total, bias,
                 C, ldc, O, btT, btB, btN, 0);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

TnqPlan tnq_plan(int nseg, int F, int R, int O, bool bt, int num_cus) {
    TnqPlan p{};
    if (nseg < 1 || nseg > kMaxM || F < 4 || F % 4 != 0 || R < 16 || R % 16 != 0) return p;
    if (O != 64 && O != 128 && O != 192) return p;
    if ((double)R * (F > 192 ? F : 192) * 4.0 >= 2147483648.0) return p;    // 2 GB descriptors on the segments and on dY (ldy <= 192)
    const int K = nseg * F;
    p.OT = O / 32;
    const bool planar_exact = F == 64 && !bt && (nseg % 3 == 0 || nseg % 2 == 0 || nseg == 1);
    if (planar_exact) {                                    // whole 64-wide planes per k-block, no padding plane
        p.planar = 1;
        p.KT = nseg % 3 == 0 ? 6 : (nseg % 2 == 0 ? 4 : 2);
        p.nkb = ceil_div(nseg, p.KT / 2);
    } else {
        // per-lane source pointers (any F % 4 == 0, batch-major or not; also 5 or 7 planes of 64, where whole-plane
        // blocks would multiply a padding plane: K = 320 is two exact blocks of 160 here).  k-block of 4 or 5 tiles per
        // wave slice (6 x 6 tiles + per-lane pointers spill): least padded K (= MFMA work), then the wider block
        if (bt && O != 192) return p;                      // (batch-major rows only occur on the x-part: 192 columns)
        int best = 5, bcost = 1 << 30, bnkb = 1;
        for (int kt = 5; kt >= 4; --kt) {
            const int nkb = ceil_div(K, 32 * kt), cost = nkb * 32 * kt;
            if (cost < bcost) { best = kt; bcost = cost; bnkb = nkb; }
        }
        p.KT = best; p.nkb = bnkb;
    }
    const int G = 2 * (num_cus > 0 ? num_cus : 256);
    int nsplit = G / p.nkb;
    if (nsplit < 1) nsplit = 1;
    int rps = round_up(ceil_div(R, nsplit), 16);
    if (rps < 64) rps = 64;
    p.rps = rps;
    p.nsplit = ceil_div(R, rps);
    p.ok = 1;
    return p;
}

namespace {
template <int KT, int OT, bool BT, bool PLANAR>
int launch_tnq_one(const TnqPlan& p, const SegPtrs& segs, int nseg, int F, int R, const float* dY, int ldy, int ycol0, int O,
                   float* partial, int btT, int btB, int btN, hipStream_t st, const char* tag) {
    constexpr int RC = 16;
    const size_t lds = 3 * (size_t)(RC * 32 * (KT + OT)) * sizeof(float);
    EEG_SET_MAX_LDS((gemm_tnq_kernel<KT, OT, RC, BT, PLANAR, false>), lds);
    EEG_LAUNCH_P(tag, (gemm_tnq_kernel<KT, OT, RC, BT, PLANAR, false>), dim3(p.nkb, p.nsplit), dim3(256), lds, st, segs, nseg, F, R, dY, ldy,
                 ycol0, O, partial, p.rps, btT, btB, btN, 0);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
template <int KT, bool BT, bool PLANAR>
int launch_tnq_ot(const TnqPlan& p, const SegPtrs& segs, int nseg, int F, int R, const float* dY, int ldy, int ycol0, int O,
                  float* partial, int btT, int btB, int btN, hipStream_t st, const char* tag) {
#define EEG_TNQ(OT) launch_tnq_one<KT, OT, BT, PLANAR>(p, segs, nseg, F, R, dY, ldy, ycol0, O, partial, btT, btB, btN, st, tag)
    if constexpr (!BT) {                                   // (batch-major segments: x-part only, 192 columns)
        if (p.OT == 2) return EEG_TNQ(2);
        if (p.OT == 4) return EEG_TNQ(4);
    }
    return p.OT == 6 ? EEG_TNQ(6) : 1;
#undef EEG_TNQ
}
}  // namespace

// hg (OT = 4) and hc (OT = 2) of one cell in one launch; both plans must agree in everything but OT (-1: not covered)
int launch_tnq_pair(const TnqPlan& pg, const TnqPlan& pc, const SegPtrs& sg, const SegPtrs& sc, int nseg, int F, int R, const float* dY, int ldy,
                    int ycol_g, int Og, float* part_g, int ycol_c, int Oc, float* part_c, hipStream_t st, const char* tag) {
    if (!pg.ok || !pc.ok || pg.OT != 4 || pc.OT != 2 || pg.KT != pc.KT || pg.planar != pc.planar || pg.nkb != pc.nkb || pg.nsplit != pc.nsplit ||
        pg.rps != pc.rps) return -1;
    constexpr int RC = 16;
    TnqJob ja{sg, ycol_g, Og, part_g}, jb{sc, ycol_c, Oc, part_c};
#define EEG_PAIR(KT, PL)                                                                                                         \
    {                                                                                                                            \
        const size_t lds = 3 * (size_t)(RC * 32 * (KT + 4)) * sizeof(float);                                                     \
        EEG_SET_MAX_LDS((gemm_tnq_pair_kernel<KT, RC, PL>), lds);                                                                \
        EEG_LAUNCH_P(tag, (gemm_tnq_pair_kernel<KT, RC, PL>), dim3(pg.nkb, 2 * pg.nsplit), dim3(256), lds, st, ja, jb, nseg, F, R, dY, ldy, \
                     pg.rps, 0);                                                                                                 \
        return hipGetLastError() == hipSuccess ? 0 : 2;                                                                          \
    }
    if (pg.planar && pg.KT == 6) EEG_PAIR(6, true)
    if (!pg.planar && pg.KT == 5) EEG_PAIR(5, false)
#undef EEG_PAIR
    return -1;
}

int launch_tnq(const TnqPlan& p, const SegPtrs& segs, int nseg, int F, int R, const float* dY, int ldy, int ycol0, int O,
               float* partial, int btT, int btB, int btN, hipStream_t st, const char* tag) {
#define EEG_TNQ(KT, BT, PL) launch_tnq_ot<KT, BT, PL>(p, segs, nseg, F, R, dY, ldy, ycol0, O, partial, btT, btB, btN, st, tag)
    if (!p.ok) return 1;
    if (p.planar) {
        if (p.KT == 2) return EEG_TNQ(2, false, true);
        if (p.KT == 4) return EEG_TNQ(4, false, true);
        return EEG_TNQ(6, false, true);
    }
    const bool bt = btT > 0;
    if (p.KT == 4) return bt ? EEG_TNQ(4, true, false) : EEG_TNQ(4, false, false);
    return bt ? EEG_TNQ(5, true, false) : EEG_TNQ(5, false, false);
#undef EEG_TNQ
}

}  // namespace eeg
