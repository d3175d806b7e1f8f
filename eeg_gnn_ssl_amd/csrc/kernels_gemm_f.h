// The hoisted GEMMs of the spectral form (spec_common.h) on v_mfma_f32_32x32x2_f32, round 6:
//   gemm_tnf_kernel  the three weight-gradient contractions of a cell in one pass over dYh          (this comment)
//   gemm_nnf_kernel  Yh_i = Xh_i Wt_i + bias with the weights of a frequency in registers            (further down)
//   gemm_dxf_kernel  dX = U [dYh_i Wt_i^T]_i: the K = 3H GEMM and the node mix back in one kernel     (16x16x4; further down)
//
// Fused weight-gradient GEMM of one cell in the eigenbasis of a shared symmetric support (spec_common.h): the three hoisted
// contractions over the rows of ONE graph frequency i
//     dWt^x_i = Xh_i^T dYh_i            (Fin x 3H)      Xh  = U^T x          (N, Sp, Fin)
//     dWt^g_i = Hh_i^T dYh_i[:, 0:2H]   (H   x 2H)      Hh  = U^T h_{t-1}    (N, Sp, H)
//     dWt^c_i = RHh_i^T dYh_i[:, 2H:3H] (H   x H)       RHh = U^T (r*h_{t-1})
// in ONE pass over dYh = U^T dXW (N, Sp, 3H): the three kernels this replaces (gemm_tnq_grouped_kernel + the two jobs of
// gemm_tnq_grouped_pair_kernel) read the 3H-wide operand twice.  Model side: the parameter gradients of
// DCGRUCell's two DiffusionGraphConv (reference model/cell.py:76-117, autograd of `torch.matmul(x, self.weight)`).
//
// Shape of the thing (H = 64): per row the operands are Fin + 64 + 64 + 192 floats and the work is 2 * 192 * (Fin + 64) FLOP --
// 24..29 FLOP/B, where the MFMA and the HBM roof of this part meet.  Construction:
//   * v_mfma_f32_32x32x2_f32: a fragment = 32 consecutive floats of an operand row per half-wave, so the LDS images are the plain
//     row-major chunks as they lie in HBM (every chunk of every operand is CONTIGUOUS there: 1-KB LDS-DMA pieces, no row maps, no
//     swizzle) and every fragment read is a conflict-free ds_read_b32 (its two 32-lane groups read one row each);
//   * a workgroup (4 waves) owns ALL output tiles of its row range: wave (e, s) = (A column half, dYh column role) accumulates
//     Xh tiles {e, e+2} x dYh tiles {3s..3s+2}, Hh tile e x gate tiles {2s, 2s+1}, RHh tile e x candidate tile 4+s: 3*NXT + 3
//     MFMAs per k-step of two rows against NXT + 6 fragment dwords -- all four waves carry the same MFMA count;
//   * ring of tnf_ns() stages of kTnfRC rows, requests one less than that many chunks ahead with counted waits on the vector-memory queue, one
//     workgroup barrier per chunk; two workgroups per CU (three for Fin <= 64);
//   * partials in the layout of the kernels it replaces ([N*spg][K][O] per problem): the fold launch is unchanged.
// Features beyond Fin in the last Xh tile are whatever follows in the image (never stored: an MFMA output row depends on its own
// A row only).  Pad rows [S, Sp) of dYh are zeros (spec_common.h), so they add nothing.
#pragma once
#include "common.h"
#include "kernels_gemm_q.h"      // IntC

namespace eeg {

#ifndef EEG_X_TNF_RC
#define EEG_X_TNF_RC 8
#endif
#ifndef EEG_X_TNF_NS
#define EEG_X_TNF_NS 5
#endif
#ifndef EEG_X_TNF_MINW
#define EEG_X_TNF_MINW 2
#endif
// narrow inputs (FXT <= 2: 96 accumulator registers) run three workgroups per CU on a three-stage ring (measured: -3 % against 2 x 5)
#ifndef EEG_X_TNF_NS2
#define EEG_X_TNF_NS2 3
#endif
#ifndef EEG_X_TNF_MINW2
#define EEG_X_TNF_MINW2 3
#endif
constexpr int kTnfRC = EEG_X_TNF_RC;
__host__ __device__ constexpr int tnf_ns(int FXT) { return FXT <= 2 ? EEG_X_TNF_NS2 : EEG_X_TNF_NS; }
__host__ __device__ constexpr int tnf_wgs_per_cu(int FXT) { return FXT <= 2 ? EEG_X_TNF_MINW2 : EEG_X_TNF_MINW; }
// floats of one stage: Xh image (FXT pieces of 256 floats) | Hh (8 x 64) | RHh (8 x 64) | dYh (8 x 192)
__host__ __device__ constexpr int tnf_stage_floats(int FXT) { return (kTnfRC / 8) * (FXT * 256 + 512 + 512 + 1536); }
__host__ __device__ constexpr size_t tnf_lds_bytes(int FXT) { return (size_t)tnf_ns(FXT) * tnf_stage_floats(FXT) * sizeof(float); }

// 1-KB requests per chunk: [Xh: FXT | Hh: 2 | RHh: 2 | dYh: 6] per 8 rows; per wave a quarter of them, rounded up
__host__ __device__ constexpr int tnf_ndma(int FXT) { return (kTnfRC / 8) * (FXT + 10); }
__host__ __device__ constexpr int tnf_per(int FXT) { return (tnf_ndma(FXT) + 3) / 4; }

// The request ring + chunk loop shared by the wave roles: `load(IntC<p>, stage, ks)` fills fragment set p for k-step ks of the stage,
// `mma(IntC<p>)` issues the role's MFMAs on fragment set p.
template <int FXT>
struct TnfRing {
    static constexpr int PER = tnf_per(FXT);
    const wbuf_t (&dsc)[PER];
    const int (&lofs)[PER];
    const unsigned (&vof)[PER];
    unsigned (&soff)[PER];
    const unsigned (&cstep)[PER];
};
template <int FXT, class LoadF, class MmaF>
__device__ __forceinline__ void tnf_ring(float* sm, const int Q, const TnfRing<FXT>& rg, LoadF load, MmaF mma) {
    constexpr int RC = kTnfRC, NS = tnf_ns(FXT), KS = RC / 2, PER = tnf_per(FXT), ST = tnf_stage_floats(FXT);
    int d_stage = 0;
    auto issue = [&]() __attribute__((always_inline)) {
        float* base = sm + d_stage * ST;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            wbuf_dma16(rg.dsc[j], base + rg.lofs[j], rg.vof[j], rg.soff[j]);
            rg.soff[j] += rg.cstep[j];
        }
        d_stage = d_stage + 1 == NS ? 0 : d_stage + 1;
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < Q) issue();
    int r_stage = 0;
    for (int q = 0; q < Q; ++q) {
        // chunk q landed (this wave's pieces: counted wait; every wave's: the barrier), and every wave is past its reads of chunk
        // q-1, whose stage takes chunk q + NS - 1
        if (q + NS - 1 <= Q) vm_wait_barrier_n<(NS - 2) * PER>();
        else EEG_VM_WAIT_BARRIER(0);
#ifndef EEG_X_TNF_NODMA
        if (q + NS - 1 < Q) issue();
#endif
        const float* st = sm + r_stage * ST;
        load(IntC<0>(), st, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks & 1) {
                if (ks + 1 < KS) load(IntC<0>(), st, ks + 1);
                EEG_SCHED_FENCE();
                mma(IntC<1>());
            } else {
                if (ks + 1 < KS) load(IntC<1>(), st, ks + 1);
                EEG_SCHED_FENCE();
                mma(IntC<0>());
            }
            EEG_SCHED_FENCE();
        }
        r_stage = r_stage + 1 == NS ? 0 : r_stage + 1;
    }
}
// D tile register v of lane (hh, l32) = (row 8*(v>>2) + 4*hh + (v&3), column l32): store a 32 x 32 tile at out[(f0 + row) * ld + c0 + col]
__device__ __forceinline__ void tnf_store_tile(float* __restrict__ out, int ld, int f0, int c0, int fmax, int lane, const f32x16& t) {
    const int hh = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int f = f0 + 8 * (v >> 2) + 4 * hh + (v & 3);
        if (f < fmax) out[(size_t)f * ld + c0 + l32] = t[v];
    }
}

template <int FXT, int S>
__device__ __forceinline__ void tnf_role(float* sm, const int Q, const int Fin, const int e, const int lane, const TnfRing<FXT>& rg,
                                         float* __restrict__ px, float* __restrict__ pg, float* __restrict__ pc) {
    constexpr int RC = kTnfRC, NXT = (FXT + 1) / 2;
    constexpr int RM = RC / 8, HI = RM * FXT * 256, RI = HI + RM * 512, YI = RI + RM * 512;
    // dYh tiles of this role: x-part {3S, 3S+1, 3S+2}; gate {2S, 2S+1}; candidate 4 + S.  Loaded: the three x tiles + one more
    // (S = 0: tile 4, the candidate's; S = 1: tile 2, the first gate tile) -- the others coincide with x tiles.
    constexpr int YX = 3 * S, YE = S == 0 ? 4 : 2;
    const int hh = lane >> 5, l32 = lane & 31;
    f32x16 ax[NXT][3], ah[2], ar;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
#pragma unroll
        for (int a = 0; a < NXT; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) ax[a][c][v] = 0.f;
        ah[0][v] = 0.f; ah[1][v] = 0.f; ar[v] = 0.f;
    }
    // fragment offsets (floats) of k-step 0 inside a stage
    const int ox = hh * Fin + 32 * e + l32, oh = HI + hh * 64 + 32 * e + l32, orr = RI + hh * 64 + 32 * e + l32, oy = YI + hh * 192 + l32;
    float fx[2][NXT], fh[2], fr[2], fy[2][4];
    auto load = [&](auto PAR, const float* st, int ks) __attribute__((always_inline)) {
        constexpr int p = decltype(PAR)::value;
#pragma unroll
        for (int a = 0; a < NXT; ++a) fx[p][a] = st[ox + 2 * ks * Fin + 64 * a];
        fh[p] = st[oh + 2 * ks * 64];
        fr[p] = st[orr + 2 * ks * 64];
#pragma unroll
        for (int c = 0; c < 3; ++c) fy[p][c] = st[oy + 2 * ks * 192 + 32 * (YX + c)];
        fy[p][3] = st[oy + 2 * ks * 192 + 32 * YE];
    };
    auto mma = [&](auto PAR) __attribute__((always_inline)) {
        constexpr int p = decltype(PAR)::value;
#pragma unroll
        for (int a = 0; a < NXT; ++a)
#pragma unroll
#ifndef EEG_X_TNF_NOMFMA
            for (int c = 0; c < 3; ++c) ax[a][c] = mfma32(fx[p][a], fy[p][c], ax[a][c]);
#else
            for (int c = 0; c < 3; ++c) ax[a][c][0] += fx[p][a] * fy[p][c];
#endif
        // gate tiles 2S, 2S+1: S = 0 -> x tiles 0, 1; S = 1 -> the extra tile (2) and x tile 0 (3)
#ifndef EEG_X_TNF_NOMFMA
        ah[0] = mfma32(fh[p], S == 0 ? fy[p][0] : fy[p][3], ah[0]);
        ah[1] = mfma32(fh[p], S == 0 ? fy[p][1] : fy[p][0], ah[1]);
        // candidate tile 4 + S: S = 0 -> the extra tile (4); S = 1 -> x tile 2 (5)
        ar = mfma32(fr[p], S == 0 ? fy[p][3] : fy[p][2], ar);
#else
        ah[0][0] += fh[p] * (S == 0 ? fy[p][0] : fy[p][3]);
        ah[1][0] += fh[p] * (S == 0 ? fy[p][1] : fy[p][0]);
        ar[0] += fr[p] * (S == 0 ? fy[p][3] : fy[p][2]);
#endif
    };
    tnf_ring<FXT>(sm, Q, rg, load, mma);
#pragma unroll
    for (int a = 0; a < NXT; ++a) {
        const int xe = e + 2 * a;
        if (xe >= FXT) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) tnf_store_tile(px, 192, 32 * xe, 32 * (YX + c), Fin, lane, ax[a][c]);
    }
    tnf_store_tile(pg, 128, 32 * e, 32 * (2 * S), 64, lane, ah[0]);
    tnf_store_tile(pg, 128, 32 * e, 32 * (2 * S + 1), 64, lane, ah[1]);
    tnf_store_tile(pc, 64, 32 * e, 32 * S, 64, lane, ar);
}

// Fin = 100 (the FFT features of a 1-s window, BASELINE's input width): three whole Xh tiles + FOUR features.  A fourth 32-wide
// tile would spend 6 of 36 MFMAs per k-step on 28 junk rows; here features 96..99 run on v_mfma_f32_4x4x1 (16 independent 4 x 4
// outer products per instruction: block b of lane group 4b..4b+3 = features 96..99 x columns 64c + 4b .. + 3, one operand ROW per
// instruction) and the whole-tile work is re-dealt so that the four waves stay level:
//   E = 0: Xh tiles 0, 2 x dYh {3S..3S+2}, Hh tile 0 x gate {2S, 2S+1}                                   8 MFMAs per k-step
//   E = 1: Xh tile 1 x dYh {3S..3S+2}, Hh tile 1 x gate {2S, 2S+1}, RHh tiles 0, 1 x candidate 4+S,     7 MFMAs
//          + the remainder of columns 0..127 (S = 0: 4 small ones per k-step) / 128..191 (S = 1: 2)
template <int S, int E>
__device__ __forceinline__ void tnf_role_r4(float* sm, const int Q, const int lane, const TnfRing<4>& rg,
                                            float* __restrict__ px, float* __restrict__ pg, float* __restrict__ pc) {
    constexpr int FXT = 4, Fin = 100, RC = kTnfRC, RM = RC / 8, HI = RM * FXT * 256, RI = HI + RM * 512, YI = RI + RM * 512;
    constexpr int YX = 3 * S, YE = S == 0 ? 4 : 2, NG = S == 0 ? 2 : 1, G0 = S == 0 ? 0 : 2;   // remainder column groups G0 .. G0+NG-1
    const int hh = lane >> 5, l32 = lane & 31;
    f32x16 t[8];                       // E = 0: [0..2] Xh0, [3..5] Xh2, [6..7] Hh0;  E = 1: [0..2] Xh1, [3..4] Hh1, [5] RHh0, [6] RHh1
    f32x4 rem[2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) t[i][v] = 0.f;
    rem[0] = rem[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int oxa = hh * Fin + (E == 0 ? 0 : 32) + l32, oxb = hh * Fin + 64 + l32;
    const int oh = HI + hh * 64 + 32 * E + l32, or0 = RI + hh * 64 + l32, oy = YI + hh * 192 + l32;
    const int oxr = 96 + (lane & 3), oyr = YI + lane;                 // remainder operands: one row per instruction
    float fa[2][4], fy[2][4], ra[2][2], rb[2][2][2];
    auto load = [&](auto PAR, const float* st, int ks) __attribute__((always_inline)) {
        constexpr int p = decltype(PAR)::value;
        fa[p][0] = st[oxa + 2 * ks * Fin];
        fa[p][1] = st[E == 0 ? oxb + 2 * ks * Fin : oh + 2 * ks * 64];           // E = 0: Xh2; E = 1: Hh1
        if (E == 0) fa[p][2] = st[oh + 2 * ks * 64];                              // Hh0
        else { fa[p][2] = st[or0 + 2 * ks * 64]; fa[p][3] = st[or0 + 32 + 2 * ks * 64]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) fy[p][c] = st[oy + 2 * ks * 192 + 32 * (YX + c)];
        if (E == 1 || S == 1) fy[p][3] = st[oy + 2 * ks * 192 + 32 * YE];
        if (E == 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                ra[p][r] = st[(2 * ks + r) * Fin + oxr];
#pragma unroll
                for (int g = 0; g < NG; ++g) rb[p][r][g] = st[oyr + (2 * ks + r) * 192 + 64 * (G0 + g)];
            }
        }
    };
    auto mma = [&](auto PAR) __attribute__((always_inline)) {
        constexpr int p = decltype(PAR)::value;
        const float g0 = S == 0 ? fy[p][0] : fy[p][3], g1 = S == 0 ? fy[p][1] : fy[p][0];   // gate tiles 2S, 2S+1
#pragma unroll
        for (int c = 0; c < 3; ++c) t[c] = mfma32(fa[p][0], fy[p][c], t[c]);
        if (E == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) t[3 + c] = mfma32(fa[p][1], fy[p][c], t[3 + c]);
            t[6] = mfma32(fa[p][2], g0, t[6]);
            t[7] = mfma32(fa[p][2], g1, t[7]);
        } else {
            const float cd = S == 0 ? fy[p][3] : fy[p][2];                         // candidate tile 4 + S
            t[3] = mfma32(fa[p][1], g0, t[3]);
            t[4] = mfma32(fa[p][1], g1, t[4]);
            t[5] = mfma32(fa[p][2], cd, t[5]);
            t[6] = mfma32(fa[p][3], cd, t[6]);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < NG; ++g) rem[g] = mfma4(ra[p][r], rb[p][r][g], rem[g]);
        }
    };
    tnf_ring<4>(sm, Q, rg, load, mma);
    if (E == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            tnf_store_tile(px, 192, 0, 32 * (YX + c), Fin, lane, t[c]);
            tnf_store_tile(px, 192, 64, 32 * (YX + c), Fin, lane, t[3 + c]);
        }
        tnf_store_tile(pg, 128, 0, 32 * (2 * S), 64, lane, t[6]);
        tnf_store_tile(pg, 128, 0, 32 * (2 * S + 1), 64, lane, t[7]);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) tnf_store_tile(px, 192, 32, 32 * (YX + c), Fin, lane, t[c]);
        tnf_store_tile(pg, 128, 32, 32 * (2 * S), 64, lane, t[3]);
        tnf_store_tile(pg, 128, 32, 32 * (2 * S + 1), 64, lane, t[4]);
        tnf_store_tile(pc, 64, 0, 32 * S, 64, lane, t[5]);
        tnf_store_tile(pc, 64, 32, 32 * S, 64, lane, t[6]);
        // remainder: register r of lane l = (feature 96 + r, column 64 (G0 + g) + l)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) px[(size_t)(96 + r) * 192 + 64 * (G0 + g) + lane] = rem[g][r];
    }
}

// grid = G * spg workgroups: workgroup y = i * spg + ls takes rows [ls * rps, min((ls+1) * rps, Sp)) of frequency i
// (rps % kTnfRC == 0, Sp % 16 == 0).  x_gstride / h_gstride: floats between two frequencies of Xh / Hh (RHh, dYh: contiguous).
template <int FXT>
__global__ __launch_bounds__(256, tnf_wgs_per_cu(FXT)) void gemm_tnf_kernel(const float* __restrict__ Xh, long long x_gstride, int Fin,
                                                         const float* __restrict__ Hh, long long h_gstride,
                                                         const float* __restrict__ RHh, const float* __restrict__ dY, int Sp, int spg, int rps,
                                                         float* __restrict__ part_x, float* __restrict__ part_g, float* __restrict__ part_c) {
    constexpr int RC = kTnfRC, RM = RC / 8, NDMA = tnf_ndma(FXT), PER = tnf_per(FXT), NX = RM * FXT, NH = RM * 2;
    constexpr int HI = NX * 256, RI = HI + NH * 256, YI = RI + NH * 256;
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), e = w & 1, s = w >> 1;
    const int y = (int)blockIdx.x, i = y / spg, ls = y - i * spg;
    const int rbeg = ls * rps;
    int rend = rbeg + rps;
    if (rend > Sp) rend = Sp;
    const int Q = rend > rbeg ? (rend - rbeg) / RC : 0;
    // this wave's PER requests per chunk: piece d = w + 4j of the list [Xh: FXT | Hh: 2 | RHh: 2 | dYh: 6] (a wave without a piece
    // of its own repeats the last one: same bytes to the same place)
    wbuf_t dsc[PER];
    int lofs[PER];
    unsigned vof[PER], soff[PER], cstep[PER];
    const wbuf_t rx = make_wbuf(Xh + (size_t)i * x_gstride), rh = make_wbuf(Hh + (size_t)i * h_gstride);
    const wbuf_t rr = make_wbuf(RHh + (size_t)i * Sp * 64), ry = make_wbuf(dY + (size_t)i * Sp * 192);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        int d = w + 4 * j;
        if (d >= NDMA) d = NDMA - 1;
        // (the piece offset rides in the per-lane part, so that a lane past the end of the Xh chunk can fall back to the chunk's start)
        if (d < NX) {                                      // the chunk is RC * Fin floats
            dsc[j] = rx; lofs[j] = d * 256; cstep[j] = (unsigned)(RC * Fin) * 4u;
            soff[j] = (unsigned)(rbeg * Fin) * 4u;
            vof[j] = d * 1024 + lane * 16 < RC * Fin * 4 ? (unsigned)(d * 1024 + lane * 16) : 0u;
        } else if (d < NX + NH) {
            dsc[j] = rh; lofs[j] = HI + (d - NX) * 256; cstep[j] = RC * 64 * 4u;
            soff[j] = (unsigned)(rbeg * 64) * 4u; vof[j] = (unsigned)((d - NX) * 1024 + lane * 16);
        } else if (d < NX + 2 * NH) {
            dsc[j] = rr; lofs[j] = RI + (d - NX - NH) * 256; cstep[j] = RC * 64 * 4u;
            soff[j] = (unsigned)(rbeg * 64) * 4u; vof[j] = (unsigned)((d - NX - NH) * 1024 + lane * 16);
        } else {
            dsc[j] = ry; lofs[j] = YI + (d - NX - 2 * NH) * 256; cstep[j] = RC * 192 * 4u;
            soff[j] = (unsigned)(rbeg * 192) * 4u; vof[j] = (unsigned)((d - NX - 2 * NH) * 1024 + lane * 16);
        }
    }
    float* px = part_x + (size_t)y * Fin * 192;
    float* pg = part_g + (size_t)y * 64 * 128;
    float* pc = part_c + (size_t)y * 64 * 64;
    const TnfRing<FXT> rg{dsc, lofs, vof, soff, cstep};
#ifndef EEG_X_TNF_NOR4
    if constexpr (FXT == 4) {
        if (Fin == 100) {                                  // three whole Xh tiles + four features (tnf_role_r4)
            if (s == 0) { if (e == 0) tnf_role_r4<0, 0>(sm, Q, lane, rg, px, pg, pc); else tnf_role_r4<0, 1>(sm, Q, lane, rg, px, pg, pc); }
            else { if (e == 0) tnf_role_r4<1, 0>(sm, Q, lane, rg, px, pg, pc); else tnf_role_r4<1, 1>(sm, Q, lane, rg, px, pg, pc); }
            return;
        }
    }
#endif
    if (s == 0) tnf_role<FXT, 0>(sm, Q, Fin, e, lane, rg, px, pg, pc);
    else tnf_role<FXT, 1>(sm, Q, Fin, e, lane, rg, px, pg, pc);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Grouped NN GEMM of the hoisted x-part in the eigenbasis, same construction: Yh_i = Xh_i Wt_i + gscale[i] * bias for every graph
// frequency i (model side: `torch.matmul(x, self.weight) + biases` of DiffusionGraphConv.forward, reference model/cell.py:113-117,
// with K = Fin instead of M * Fin, spec_common.h).  K is small (64 / 100) and the output wide (3H = 192): per 64 rows 16..26 KB come
// in, 48 KB go out, 1.6..2.5 MFLOP are spent -- the HBM roof of the part is the tighter one.  Construction:
//   * the weights of a frequency live in REGISTERS (wave (rw, cw): row tile rw of the 64-row chunk x column tiles 3cw..3cw+2; its
//     fragments W[8q + 4hh + s][32(3cw+t) + c] are 12 * KQ dwords, loaded once per frequency a workgroup visits): the loop has no
//     weight traffic at all, LDS holds only the ring of activation chunks;
//   * an activation fragment is ONE ds_read_b128 per 4 k-steps (lane (row, hh): the 16 bytes k = 8q + 4hh .. +3 of its row).  A row of
//     K = 100 floats is 25 sixteen-byte units -- odd, so the 16 rows of a ds_read_b128 lane group fall on 16 different slots of the
//     plain row-major image (contiguous 1-KB LDS-DMA pieces); K = 64 (16 units) is XOR-swizzled in the SOURCE addresses of the DMA;
//   * non-transposed issue: lane (c, hh) holds column c of 16 rows, every store instruction writes two whole 128-byte lines;
//   * persistent workgroups (2 per CU) over a balanced contiguous range of 64-row chunks; a range may cross into the next frequency
//     (weights reloaded there); ragged ends (Sp % 64) through bounded descriptors: no per-lane guards anywhere.
// KQ = ceil(K / 8); SWZ: K == 64.  A: (G, Sp, K) with a_gstride floats between frequencies; Wr: SpecPack::sxr; C: (G, Sp, 192).
#ifndef EEG_X_NNF_NS
#define EEG_X_NNF_NS 3
#endif
constexpr int kNnfNS = EEG_X_NNF_NS;
__host__ __device__ constexpr int nnf_stage_floats(int K) { return 64 * K + 4; }          // + one zeroed 16-byte unit behind the last row
__host__ __device__ constexpr size_t nnf_lds_bytes(int K) { return (size_t)kNnfNS * nnf_stage_floats(K) * sizeof(float); }

template <int KQ, bool SWZ>
__global__ __launch_bounds__(256, 2) void gemm_nnf_kernel(const float* __restrict__ A, long long a_gstride, int K, int Sp, int G,
                                                         const float* __restrict__ Wr, unsigned w_gstride, float* __restrict__ C,
                                                         const float* __restrict__ bias, const float* __restrict__ gscale) {
    constexpr int NS = kNnfNS;
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), rw = w & 1, cw = w >> 1, hh = lane >> 5, l32 = lane & 31;
    const int ST = nnf_stage_floats(K), CPG = ceil_div(Sp, 64), total = G * CPG;
    const int NPC = ceil_div(64 * K, 256), PER = ceil_div(NPC, 4);        // 1-KB pieces of a chunk; per wave (the last one repeated)
    const int g0 = (int)(((long long)blockIdx.x * total) / gridDim.x), g1 = (int)(((long long)(blockIdx.x + 1) * total) / gridDim.x);
    const int Q = g1 - g0;
    if (tid < 4 * NS) sm[(tid >> 2) * ST + 64 * K + (tid & 3)] = 0.f;     // the unit behind the last row of every stage
    // ---- request side -------------------------------------------------------------------------------------------------------------
    // piece p = w + 4j covers bytes [1024 p, 1024 p + 1024) of the chunk image; lane part of the source: SWZ: row (l >> 4) of the
    // piece's 4 rows, unit (l & 15) ^ ((4w + (l >> 4)) & 15) (the row index mod 16 is the same for every j); else the image is the chunk
    const unsigned lane_src = SWZ ? (unsigned)((lane >> 4) * 256 + (((lane & 15) ^ ((4 * w + (lane >> 4)) & 15)) << 4)) : (unsigned)lane * 16u;
    int d_g = g0, d_stage = 0, d_grp = -1;
    wbuf_t ra = make_wbuf_n(A, 0u);
    auto issue = [&]() __attribute__((always_inline)) {
        const int grp = d_g / CPG, lc = d_g - grp * CPG;
        if (grp != d_grp) { d_grp = grp; ra = make_wbuf_n(A + (size_t)grp * a_gstride, (unsigned)(Sp * K) * 4u); }
        float* base = sm + d_stage * ST;
        const unsigned cb = (unsigned)(lc * 64 * K) * 4u;
        for (int j = 0; j < PER; ++j) {
            int p = w + 4 * j;
            if (p >= NPC) p = NPC - 1;
            wbuf_dma16(ra, base + p * 256, cb + (unsigned)p * 1024u + lane_src, 0u);
        }
        d_stage = d_stage + 1 == NS ? 0 : d_stage + 1;
        ++d_g;
    };
    // ---- compute side -------------------------------------------------------------------------------------------------------------
    float wr[KQ][4][3];
    f32x16 acc[3];
    float binit[3] = {0.f, 0.f, 0.f};
    int c_grp = -1;
    wbuf_t rc = make_wbuf_n(C, 0u);
    const int row = 32 * rw + l32;                                        // this lane's activation row inside a chunk
    const int a_row = row * K;
    const int ccol = 32 * 3 * cw + l32;                                   // first of this lane's three output columns
#pragma unroll 1
    for (int p = 0; p < NS - 1; ++p)
        if (p < Q) issue();
    int r_stage = 0;
#pragma unroll 1
    for (int q = 0; q < Q; ++q) {
        const int g = g0 + q, grp = g / CPG, lc = g - grp * CPG;
        if (grp != c_grp) {                                               // a new frequency: its weights, bias scale and output window
            c_grp = grp;
            const wbuf_t rwt = make_wbuf(Wr + (size_t)grp * w_gstride);
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < 3; ++t) wr[kq][s][t] = wbuf_ld(rwt, (unsigned)(4 * hh * 192 + ccol), (unsigned)((8 * kq + s) * 192 + 32 * t));
            const float gs = bias != nullptr ? gscale[grp] : 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t) binit[t] = bias != nullptr ? gs * bias[ccol + 32 * t] : 0.f;
            rc = make_wbuf_n(C + (size_t)grp * Sp * 192, (unsigned)(Sp * 192) * 4u);
        }
        // Queue discipline (it retires in order, stores included): behind the requests of chunk q the queue holds the requests of the
        // chunks q+1 .. q+NS-2 that exist (PER each) and the 48-store bursts of the chunks q-NS+1 .. q-1 that exist: a counted wait for
        // the largest supported count not above that (the counter has 6 bits).
        {
            int nd = Q - 1 - q;
            if (nd > NS - 2) nd = NS - 2;
            const int n = 48 * (q < NS - 1 ? q : NS - 1) + PER * nd;
            if (n >= 63) vm_wait_n<63>(); else if (n >= 55) vm_wait_n<55>(); else if (n >= 52) vm_wait_n<52>(); else if (n >= 48) vm_wait_n<48>();
            else if (n >= 14) vm_wait_n<14>(); else if (n >= 8) vm_wait_n<8>(); else if (n >= 7) vm_wait_n<7>(); else if (n >= 4) vm_wait_n<4>();
            else vm_wait_n<0>();
        }
#ifndef EEG_X_NNF_NOBAR
        EEG_LDS_BARRIER();                                                // chunk q landed in every wave; all are past their reads of chunk q-1
#endif
#ifndef EEG_X_NNF_NODMA
        if (q + NS - 1 < Q) issue();
#endif
        const float* st = sm + r_stage * ST;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = binit[t];
        f32x4 a4 = *reinterpret_cast<const f32x4*>(st + a_row + (SWZ ? ((hh ^ (row & 15)) << 2) : 4 * hh));
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            f32x4 an = a4;
#ifndef EEG_X_NNF_NOLDS
            if (kq + 1 < KQ) an = *reinterpret_cast<const f32x4*>(st + a_row + (SWZ ? (((2 * (kq + 1) + hh) ^ (row & 15)) << 2) : 4 * (2 * (kq + 1) + hh)));
#endif
            EEG_SCHED_FENCE();
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = mfma32(a4[s], wr[kq][s][t], acc[t]);
            EEG_SCHED_FENCE();
            a4 = an;
        }
        // D tile register v of lane (hh, c) = (row 8*(v>>2) + 4*hh + (v&3), column c)
        // (one per-lane offset per 8-row band; the rest of a store's address fits the instruction's 12-bit offset)
        unsigned ob[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            ob[b] = (unsigned)((lc * 64 + 32 * rw + 4 * hh + 8 * b) * 192 + ccol);
            EEG_PIN(ob[b]);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
#ifndef EEG_X_NNF_NOSTORE
            for (int v = 0; v < 16; ++v) wbuf_st1(rc, ob[v >> 2] + (unsigned)((v & 3) * 192 + 32 * t), 0u, acc[t][v]);
#else
            for (int v = 0; v < 16; ++v) { EEG_USE(acc[t][v]); (void)rc; }
#endif
        EEG_SCHED_FENCE();                                                // (a store's data registers must not be rewritten right behind it)
        r_stage = r_stage + 1 == NS ? 0 : r_stage + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Input gradient of a spectral layer in ONE kernel: dX[s][j] = sum_i U[j][i] (dYh_i[s] Wt_i^T) -- the grouped NN GEMM over K = 3H
// (gemm_nng_kernel<1>: dXh, node-major) and the node mix back (spec_mix: dX, sample-major) without the round trip of dXh through
// HBM.  Model side: autograd's gradient of `torch.matmul(x, self.weight)` w.r.t. x and of the diffusion in front of it (reference
// model/cell.py:76-117) for layers above the first.  A workgroup takes 32 rows (samples) and walks ALL N frequencies:
//   * wave w = column tile w of the 64 output features, two 16-row tiles; per frequency 12 chunks of 16 k: activation fragments by
//     ds_read_b128 from a ring of [32 rows x 192] images (XOR-swizzled in the SOURCE addresses of the LDS-DMA: a 768-byte row is 48
//     sixteen-byte units -- even -- so the plain image would put the 16 rows of a lane group on one slot), weight fragments as the
//     lane's own float4 of the quad pack (SpecPack::sxtq) straight from L2 into registers, re-requested for the NEXT frequency
//     right behind their last use (one register set);
//   * transposed issue: a lane ends up with 4 consecutive features of one row; the frequency's tile T is folded into the N node
//     accumulators on the VALU (acc_j += U[j][i] * T: 8 N FMAs per 96 MFMAs), so nothing but dX leaves the kernel;
//   * rows past S / Sp through bounded descriptors (dropped stores / zero-filled requests).
// N <= kDxfMaxN (accumulators: 8 N registers).
constexpr int kDxfMaxN = 20, kDxfNS = 3, kDxfRows = 32, kDxfStage = kDxfRows * 192;
__host__ __device__ constexpr size_t dxf_lds_bytes() { return (size_t)kDxfNS * kDxfStage * sizeof(float); }

// NT: the node count as a literal (19: the EEG montage -- no branches in the fold), 0 = the runtime argument.
template <int NT>
__global__ __launch_bounds__(256, 2) void gemm_dxf_kernel(const float* __restrict__ dYh, int Sp, int S, int Nrt, const float* __restrict__ Wtq,
                                                         unsigned w_gstride, const float* __restrict__ basis, float* __restrict__ dX) {
    const int N = NT > 0 ? NT : Nrt;
    constexpr int NS = kDxfNS, ST = kDxfStage, RB = kDxfRows, PER = 6;        // 24 one-KB pieces per image, 6 per wave
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), lr = lane & 15, lg = lane >> 4;
    const int r0 = (int)blockIdx.x * RB;
    // ---- request side: piece p = w + 4j, image position P = 64 p + lane (16-byte units): row P / 48, slot P % 48 holds unit
    //      (slot & ~15) | ((slot & 15) ^ (row & 15)) of that row
    unsigned src[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int P = 64 * (w + 4 * j) + lane, row = P / 48, slot = P - row * 48;
        const int unit = (slot & ~15) | ((slot & 15) ^ (row & 15));
        src[j] = (unsigned)((r0 + row) * 768 + unit * 16);
    }
    // Every vector-memory operation of the frequency loop is UNCONDITIONAL (past the last frequency the last image / weight set is
    // requested again): with a fixed sequence per iteration the compiler's counted waits in front of the weight registers come out
    // exact (vmcnt(17)); with conditional requests it drained the queue -- the image requested a moment earlier included -- in every
    // iteration (measured: 0.098 ms against 0.0xx).
    int d_i = 0, d_stage = 0;
    auto issue = [&]() __attribute__((always_inline)) {
        const wbuf_t ra = make_wbuf_n(dYh + (size_t)d_i * Sp * 192, (unsigned)(Sp * 192) * 4u);
        float* base = sm + d_stage * ST;
#pragma unroll
        for (int j = 0; j < PER; ++j) wbuf_dma16(ra, base + (w + 4 * j) * 256, src[j], 0u);
        d_stage = d_stage + 1 == NS ? 0 : d_stage + 1;
        if (d_i + 1 < N) ++d_i;
    };
    // ---- compute side -----------------------------------------------------------------------------------------------------------
    f32x4 acc[kDxfMaxN][2];
#pragma unroll
    for (int j = 0; j < kDxfMaxN; ++j) acc[j][0] = acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 wq[12];
    const unsigned wl = (unsigned)(w * 64 + lane) * 4u;                          // this lane's float4 of column tile w inside a chunk
    static_assert(kDxfNS == 3, "the queue discipline below is written for two images in flight");
    issue();
    issue();
    EEG_SCHED_FENCE();                                     // (the queue order below is what the counted wait is written for)
    {
        const wbuf_t rw0 = make_wbuf(Wtq);
#pragma unroll
        for (int c = 0; c < 12; ++c) wq[c] = wbuf_ld4(rw0, wl, (unsigned)(c * 4 * 256));
    }
    // queue, oldest first: image 0 | image 1 | the 12 weight requests of frequency 0.  From here on image i has landed by the time
    // the weights of frequency i-1 were consumed (the queue retires in order and they were requested behind it): no wait in the loop.
    vm_wait_n<PER + 12>();
    int r_stage = 0;
    const int a_row = lr * 192;                            // fragments: unit 4c + lg of row 16t + lr, swizzled
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
        EEG_LDS_BARRIER();                                 // image i landed in every wave; all are past their reads of image i-1
#ifndef EEG_X_DXF_NODMA
        issue();                                           // image i + 2 into the stage of image i - 1
#endif
        const wbuf_t rwn = make_wbuf(Wtq + (size_t)(i + 1 < N ? i + 1 : i) * w_gstride);
        const float* st = sm + r_stage * ST;
        // U[j][i], j = 0 .. N-1: scalar loads issued here, consumed behind the MFMAs
        float uu[kDxfMaxN];
#pragma unroll
        for (int j = 0; j < kDxfMaxN; ++j) uu[j] = (NT > 0 ? j < NT : j < N) ? basis[j * N + i] : 0.f;
        f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
        auto frag_off = [&](int c) __attribute__((always_inline)) {
            const int u = 4 * c + lg;
            return a_row + (((u & ~15) | ((u & 15) ^ lr)) << 2);
        };
        f32x4 fa[2][2];
        fa[0][0] = *reinterpret_cast<const f32x4*>(st + frag_off(0));
        fa[0][1] = *reinterpret_cast<const f32x4*>(st + 16 * 192 + frag_off(0));
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            if (c + 1 < 12) {                              // the next chunk's fragments ahead of this chunk's MFMAs
                fa[(c + 1) & 1][0] = *reinterpret_cast<const f32x4*>(st + frag_off(c + 1));
                fa[(c + 1) & 1][1] = *reinterpret_cast<const f32x4*>(st + 16 * 192 + frag_off(c + 1));
            }
            EEG_SCHED_FENCE();
#pragma unroll
            for (int e = 0; e < 4; ++e) {                  // transposed issue: D[feature 4 lg + v][row lr]
                t0 = mfma16(wq[c][e], fa[c & 1][0][e], t0);
                t1 = mfma16(wq[c][e], fa[c & 1][1][e], t1);
            }
            EEG_SCHED_FENCE();
#ifndef EEG_X_DXF_NOW
            wq[c] = wbuf_ld4(rwn, wl, (unsigned)(c * 4 * 256));   // the next frequency's fragments into the registers just consumed
#endif
            EEG_SCHED_FENCE();
        }
#pragma unroll
        for (int j = 0; j < kDxfMaxN; ++j)
            if (NT > 0 ? j < NT : j < N) {
                acc[j][0] += t0 * uu[j];
                acc[j][1] += t1 * uu[j];
            }
        r_stage = r_stage + 1 == NS ? 0 : r_stage + 1;
    }
    // ---- dX[(r * N + j) * 64 + 16 w + 4 lg .. + 3], r = r0 + 16 t + lr; rows >= S fall off the end of the descriptor
    const wbuf_t rx = make_wbuf_n(dX, (unsigned)(S * N * 64) * 4u);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const unsigned rb = (unsigned)((r0 + 16 * t + lr) * N * 64 + 16 * w + 4 * lg);
#pragma unroll
        for (int j = 0; j < kDxfMaxN; ++j)
            if (j < N) wbuf_st4(rx, rb + (unsigned)(j * 64), 0u, acc[j][t]);
    }
}

}  // namespace eeg
