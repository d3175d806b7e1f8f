// Persistent DCGRU sequence kernels (the recurrent half of model/cell.py:182-210 driven by the
// time loop of model/model.py:90-96), one launch per layer and direction.
//
// Design (MI355X-first):
//  * samples are independent, so ONE WORKGROUP OWNS ONE CLIP for the whole sequence: no
//    inter-workgroup traffic, no grid barrier; B workgroups fill the 256 CUs at B = 256.
//  * the recurrent weights (W^h, M*H x 3H fp32 = 147..246 kB) do not fit LDS, so every wave keeps
//    its column slice as MFMA fragments IN REGISTERS for all T steps, next to the fragments of the
//    clip's (M-1) hop-polynomial matrices (they are constant over the sequence too).
//  * the hidden state / gradient tile and its M hop-diffused copies stay in LDS.
//  * every MFMA is issued "transposed": D^T(16 cols x 16 nodes) = W^T-frag (A operand) x
//    X^T-frag (B operand).  The fragment lane maps are identical to the untransposed product, but
//    the result lands as lane (node = lane&15, 4 CONSECUTIVE columns 4*(lane>>4)..+3), so every
//    epilogue access — XW / saved-gate loads, r,u,c,h stores, LDS tile updates — is one 16-byte
//    vector op per (lane, tile) with a single validity guard (node < N).
//  * node-feature fragments come from LDS as ds_read_b128 (four k per lane, K order permuted to
//    match: common.h kperm), fetched one quad ahead of the MFMAs that consume them.
//  * the input half of the diffusion convolution (x-part, + biases) is hoisted out of the
//    recurrence (kernels_gemm.h) and arrives as XW (T,B,N,3H) = [r | u | c] pre-activations.
//
// fwd per step:  hops(h) -> G = XW_g + hops(h) Wg^h -> r,u = sigmoid -> hops(r*h)
//                -> C = XW_c + hops(r*h) Wc^h -> c = act(C) -> h' = u*h + (1-u)*c
// bwd per step:  SURVEY.md §9 "Cell backward" with P_m^T adjoint mixes; emits dXW = [dR|dU|dC]
//                per step (consumed afterwards by the hoisted weight-gradient / dX GEMMs).
//
// Ownership: for column tile ct and node tile nt in {0,1} a lane owns node n = 16*nt + (lane&15)
// and columns 16*ct + 4*(lane>>4) + 0..3.  Nodes >= N are padding: computed (finite values that
// never leave padding rows), written to LDS as zeros, never stored to HBM.
#pragma once
#include "common.h"
#include "lds_diffuse.h"

namespace eeg {

template <int H, int M>
struct SeqGeom {
    static constexpr int KA = M * H, KAP = lds_stride_x(KA), KS = KA / 4;        // h-wide hop tile (swizzled, common.h)
    static constexpr int KG = M * 2 * H, KGP = lds_stride_x(KG), KSG = KG / 4;   // 2H-wide hop tile (bwd)
    static constexpr int NGT = 2 * H / 16, NCT = H / 16;                         // gate / cand col tiles
    static constexpr int GT = ceil_div(NGT, 4), CT = ceil_div(NCT, 4);           // per wave (4 waves)
    static constexpr size_t fwd_lds_floats() { return (size_t)(M - 1) * kPFloats + 2 * 32 * KAP; }
    // Node rows of the backward kernel's LDS tiles: 32 (two MFMA node tiles), or -- where 32 rows exceed the
    // 160 KB of a CU (H=64, M=7) and the montage has at most 20 nodes, so that the second tile runs on the
    // 4x4x1 MFMA and only rows 16..19 are ever read -- 20.
    static constexpr size_t bwd_lds_floats(int rows) { return (size_t)(M - 1) * kPFloats + (size_t)rows * (KAP + KGP); }
    static constexpr int bwd_rows(int nks) { return (nks == 5 && bwd_lds_floats(32) * sizeof(float) > kMaxLdsBytes) ? 20 : 32; }
};


// acc[i][nt] += W-frag[i][.] x X(32 nodes x 4*NKS, LDS, stride)^T for NT column tiles; the node
// fragments are read as float4 (k = 16q + 4*(lane>>4) + j) one quad ahead of their use.
//
// REM4 (montages with at most 20 nodes): the second node tile holds at most 4 real nodes, so instead
// of a second 16x16x4 stream (100 % extra matrix work for 3 nodes) it is computed with
// v_mfma_f32_4x4x1 (16 independent 4x4 outer products per instruction = 25 % extra): block = (lane
// group lg, column quad), B operand = the SAME weight register, A operand = X[node 16 + (lane&3)][k];
// register r of a lane is then the lane group's partial of out[node 16 + r][col lr].  The partials are
// reduce-scattered in registers (L1 below: common.h rem4_reduce), one element of the 4 x 16 remainder tile per lane.
// MODE (REM4 only): 0 = both node tiles; 1 = only the 16-node tile (acc[.][0]); 2 = only the 4x4x1 remainder
// (acc[.][1]) -- the two-wave forward kernel splits a GEMM between its waves that way.
// QM (the two-role BPTT kernel splits its K = M*2H gate GEMM by column half): 0: weight quad q' reads tile quad q';
// 1 / 2: the quads of the dR / dU halves of every 2H-wide hop slot (tile quad 8*(q'/4) + q'%4 [+ 4]).
// WQ: `w` holds ALL k-steps and is indexed by the mapped quad too (only half of the quads are visited).
// L1 (REM4 only): the remainder leaves in the ONE-VALUE-PER-LANE layout -- lane (lr, lg) <-> node 16 + lg, column lr of the tile --
// as acc[i][1][0] += out[16 + lg][lr] (components 1..3 of acc[i][1] are not touched): the four lane-group partials are reduced
// and scattered with three register swaps (common.h rem4_reduce), no LDS hand-over, and the remainder epilogue of the caller
// is scalar work on 64 distinct elements instead of float4 work on lanes of which a quarter hold real nodes.
// compile-time loop: f(SeqIdx<I>()) for I in [B, E) -- every index is a constant inside the body (register arrays stay registers)
template <int V> struct SeqIdx { static constexpr int value = V; };
template <int B, int E, typename Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    if constexpr (B < E) { f(SeqIdx<B>()); static_for<B + 1, E>(f); }
}
#ifndef EEG_REM_COVER
#define EEG_REM_COVER 128
#endif
// TWOPASS item stream (N16 quads of 16x16x4 MFMAs, then N4 quads of 4x4x1 MFMAs): index of the last item whose LDS fragment is
// requested before item i starts -- the MFMAs of the items in between (128 / 32 cycles per quad) cover the ds_read_b128 latency.
template <int N16, int N4>
constexpr int rem_ahead(int i) {
    int cyc = 0, l = i;
    while (l + 1 < N16 + N4 && cyc < EEG_REM_COVER) { cyc += l < N16 ? 128 : 32; ++l; }
    return l;
}

// QB (TWOPASS only): the stream starts at quad QB -- the quads before it were run earlier by mfma_tile_quads, whose chain state comes
// in through `carry` (carry[0..3] = the four 4x4x1 chains, carry[4] = the second 16x16x4 chain; acc[0][0] holds the first).
template <int NT, int NKS, bool REM4, int MODE = 0, int QM = 0, bool WQ = false, bool TWOPASS = false, bool L1 = false, int QB = 0>
__device__ __forceinline__ void mfma_nodes32(const float* __restrict__ X, int stride, int lane, int lr, int lg,
                                             const float (&w)[NT][NKS], f32x4 (&acc)[NT][2], const f32x4* carry = nullptr) {
    static_assert(NKS % 4 == 0, "K must be a multiple of 16");
    static_assert(MODE == 0 || REM4, "split modes exist for the 4x4x1 remainder only");
    static_assert(!L1 || REM4, "the one-value-per-lane remainder layout exists for the 4x4x1 remainder only");
    static_assert(!(REM4 && MODE != 1) || L1, "the 4x4x1 remainder always leaves in the one-value-per-lane layout");
    constexpr bool DO16 = MODE != 2, DO4 = REM4 && MODE != 1;
    // swizzled tile (common.h): quad q of row r is the 16-byte piece 16*(q>>2) + ((4*(q&3) + lg) ^ sigma4(r))
    const int s0 = lg ^ sigma4(lr), s1 = REM4 ? (lg ^ sigma4(lane & 3)) : s0;
    const float* p0 = X + lr * stride;
    const float* p1 = X + (REM4 ? 16 + (lane & 3) : 16 + lr) * stride;
    constexpr int NQ = WQ ? NKS / 8 : NKS / 4;                       // quads visited
    constexpr auto qmap = [](int qw) constexpr { return QM == 0 ? qw : 8 * (qw >> 2) + (qw & 3) + (QM == 2 ? 4 : 0); };
    auto frag = [&](const float* rowp, int sx, int qw) {
        const int q = qmap(qw);
        return *reinterpret_cast<const float4*>(rowp + 64 * (q >> 2) + 4 * ((4 * (q & 3)) ^ sx));
    };
    f32x4 rem[NT][4];           // one chain per k-step of the quad: consecutive 4x4x1 MFMAs are independent
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rem[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A lone accumulator makes every 16x16x4 MFMA wait for the one before it: 52 cycles per MFMA instead of the 32 of the issue
    // rate, 64 beside a second MFMA wave on the SIMD (tools/micro/chain_lab.hip).  With one column tile and the second node tile
    // on the 4x4x1 path the k-steps therefore alternate between two chains, added up at the end.
    constexpr bool SPLIT = DO16 && REM4 && NT == 1;
    f32x4 alt[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) alt[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (TWOPASS) {
        // A wave that has the matrix pipe to itself (role B's dR half in window 2 of the BPTT kernel): all 16x16x4 MFMAs of
        // the GEMM, then all its 4x4x1 MFMAs -- one change of shape (~43 cycles) instead of one per quad.  It pays in the
        // two-wave kernels only: in the single-wave kernels (one or two column tiles per wave, M = 5) the same stream measured
        // 3 % (forward) to 13-30 % (BPTT) SLOWER than the quad-interleaved order (profiles/r04_h_singlewave_twopass.txt).
        static_assert(REM4 && NT == 1, "two-pass stream: one column tile, 4x4x1 remainder");
        constexpr int N16 = DO16 ? NQ - QB : 0, N4 = DO4 ? NQ - QB : 0, NI = N16 + N4;
        if (QB > 0 && carry != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rem[0][j] = carry[j];
            alt[0] = carry[4];
        }
        float4 xb[NI];
        static_for<0, rem_ahead<N16, N4>(0) + 1>([&](auto L) __attribute__((always_inline)) {
            constexpr int l = decltype(L)::value;
            xb[l] = l < N16 ? frag(p0, s0, QB + l) : frag(p1, s1, QB + l - N16);
        });
        static_for<0, NI>([&](auto IT) __attribute__((always_inline)) {
            constexpr int it = decltype(IT)::value;
            constexpr int lo = it > 0 ? rem_ahead<N16, N4>(it > 0 ? it - 1 : 0) + 1 : 0, hi = it > 0 ? rem_ahead<N16, N4>(it) + 1 : 0;
            static_for<lo, hi>([&](auto L) __attribute__((always_inline)) {
                constexpr int l = decltype(L)::value;
                xb[l] = l < N16 ? frag(p0, s0, QB + l) : frag(p1, s1, QB + l - N16);
            });
            EEG_SCHED_FENCE();
            constexpr int q = QB + (it < N16 ? it : it - N16);
            constexpr int qw4 = 4 * (WQ ? qmap(q) : q);
            const float x[4] = {xb[it].x, xb[it].y, xb[it].z, xb[it].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (it < N16) {
                    if (j & 1) alt[0] = mfma16(w[0][qw4 + j], x[j], alt[0]);
                    else acc[0][0] = mfma16(w[0][qw4 + j], x[j], acc[0][0]);
                } else {
                    rem[0][j] = mfma4(x[j], w[0][qw4 + j], rem[0][j]);
                }
            }
            if (it < N16) EEG_PIN(alt[0]);       // (else LLVM sinks the whole second chain behind the 4x4x1 pass, as ONE dependent chain)
            EEG_SCHED_FENCE();
        });
    }
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (!TWOPASS && DO16) a0 = frag(p0, s0, 0);
    if (!TWOPASS && (!REM4 || DO4)) a1 = frag(p1, s1, 0);
#pragma unroll
    for (int q = 0; q < (TWOPASS ? 0 : NQ); ++q) {
        const int qw4 = 4 * (WQ ? qmap(q) : q);                      // first weight k-step of this quad
        float4 n0 = a0, n1 = a1;
        if (q + 1 < NQ) {
            if (DO16) n0 = frag(p0, s0, q + 1);
            if (!REM4 || DO4) n1 = frag(p1, s1, q + 1);
        }
        EEG_SCHED_FENCE();      // next quad's fragments are in flight while this quad's MFMAs issue
        const float x0[4] = {a0.x, a0.y, a0.z, a0.w}, x1[4] = {a1.x, a1.y, a1.z, a1.w};
        if (DO16) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    if (SPLIT && (j & 1)) alt[i] = mfma16(w[i][qw4 + j], x0[j], alt[i]);
                    else acc[i][0] = mfma16(w[i][qw4 + j], x0[j], acc[i][0]);
                    if (!REM4) acc[i][1] = mfma16(w[i][qw4 + j], x1[j], acc[i][1]);
                }
        }
        if (DO4) {                  // the quad's 4x4x1 MFMAs as one group (fewer switches between MFMA shapes)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i) rem[i][j] = mfma4(x1[j], w[i][qw4 + j], rem[i][j]);
        }
        EEG_SCHED_FENCE();
        a0 = n0;
        a1 = n1;
    }
    if (SPLIT) {
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i][0] += alt[i];
    }
    if constexpr (DO4 && L1) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            f32x4 t = (rem[i][0] + rem[i][1]) + (rem[i][2] + rem[i][3]);
            EEG_PIN(t);                                  // (keeps the three adds packed)
            acc[i][1][0] += rem4_reduce(t);
        }
    }
}

// Quads [Q0, Q1) of a one-column-tile GEMM over a swizzled tile with the accumulator chains held by the caller: acc / alt = the two
// 16x16x4 chains of nodes 0..15, rem[4] = the four 4x4x1 chains of the remainder (not reduced here).  Lets a wave run a slice
// of a GEMM in one window and the rest in another (the hop-0 slot of the next step's update gate: seq_fwd2_kernel, role B).
template <int NKS, int Q0, int Q1>
__device__ __forceinline__ void mfma_tile_quads(const float* __restrict__ X, int stride, int lane, int lr, int lg, const float (&w)[NKS],
                                                f32x4& acc, f32x4& alt, f32x4 (&rem)[4]) {
    const int s0 = lg ^ sigma4(lr), s1 = lg ^ sigma4(lane & 3);
    const float* p0 = X + lr * stride;
    const float* p1 = X + (16 + (lane & 3)) * stride;
    auto frag = [&](const float* rowp, int sx, int q) {
        return *reinterpret_cast<const float4*>(rowp + 64 * (q >> 2) + 4 * ((4 * (q & 3)) ^ sx));
    };
    if constexpr (Q0 >= Q1) return;
    float4 a0 = frag(p0, s0, Q0), a1 = frag(p1, s1, Q0);
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        float4 n0 = a0, n1 = a1;
        if (q + 1 < Q1) { n0 = frag(p0, s0, q + 1); n1 = frag(p1, s1, q + 1); }
        EEG_SCHED_FENCE();
        const float x0[4] = {a0.x, a0.y, a0.z, a0.w}, x1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j & 1) alt = mfma16(w[4 * q + j], x0[j], alt);
            else acc = mfma16(w[4 * q + j], x0[j], acc);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) rem[j] = mfma4(x1[j], w[4 * q + j], rem[j]);
        EEG_SCHED_FENCE();
        a0 = n0;
        a1 = n1;
    }
}

// Spectral form (spec_common.h): U^T applied to NG 16-column tiles that live in swizzled LDS node-row tiles, results straight to the
// NODE-major global tensor (N, Sp, ld): out[i][row][gcol[g] + col] = sum_n U[n][i] tile_g[n][src_col[g] + col].  Issued transposed
// like every node mix here (features as A operand): frequencies 0..15 come out of one 16x16x4 chain per tile (lane (lr, lg): four
// consecutive columns of frequency lr), frequencies 16..19 of a 4x4x1 chain (A operand = U[n][16 + (lane & 3)], B operand = the
// same feature value; rem4_reduce leaves ONE element per lane: frequency 16 + lg, column lr).  uf / u4: this lane's U fragments
// (load_spec_frags), zero for nodes / frequencies >= N.  The NG chains are independent (no lone-accumulator stalls).
template <int NKS>
__device__ __forceinline__ void load_spec_frags(const float* __restrict__ U, int N, int lane, float (&uf)[NKS], float (&u4)[NKS]) {
    const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int n = 4 * ks + lg, i4 = 16 + (lane & 3);
        uf[ks] = (n < N && lr < N) ? U[n * N + lr] : 0.f;
        u4[ks] = (n < N && i4 < N) ? U[n * N + i4] : 0.f;
    }
}
template <int NKS, int NG>
__device__ __forceinline__ void spec_mix_tiles_out(const float* const (&tile)[NG], const int (&stride)[NG], const int (&src_col)[NG],
                                                   const int (&gcol)[NG], const float (&uf)[NKS], const float (&u4)[NKS], int lr, int lg,
                                                   wbuf_t out, unsigned voff0, unsigned voff1, unsigned soff, bool valid0, bool valid1) {
    float a[NG][NKS];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) a[g][ks] = tile[g][lds_sw(4 * ks + lg, src_col[g] + lr, stride[g])];
    f32x4 acc[NG], alt[NG], rem[NG], rem2[NG];                                // (NG == 1: two 16x16x4 chains -- a lone accumulator
#pragma unroll                                                                //  makes every MFMA wait for the one before it)
    for (int g = 0; g < NG; ++g) acc[g] = alt[g] = rem[g] = rem2[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (NG == 1 && (ks & 1)) alt[g] = mfma16(a[g][ks], uf[ks], alt[g]);
            else acc[g] = mfma16(a[g][ks], uf[ks], acc[g]);
        }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (ks & 1) rem2[g] = mfma4(u4[ks], a[g][ks], rem2[g]);
            else rem[g] = mfma4(u4[ks], a[g][ks], rem[g]);
        }
    if (NG == 1) acc[0] += alt[0];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        // (the register swaps of rem4_reduce are inline asm, which the compiler's hazard recognizer does not look into: they must not be
        //  the first readers of an MFMA result -- the two-chain sum is an ordinary VALU instruction and gets the matrix-write wait states)
        f32x4 tr = rem[g] + rem2[g];
        EEG_PIN(tr);
        const float r1 = rem4_reduce(tr);
        if (valid0) wbuf_st4(out, voff0 + (unsigned)gcol[g], soff, acc[g]);
        if (valid1) wbuf_st1(out, voff1 + (unsigned)gcol[g], soff, r1);
    }
}

// The 4-node remainder of a one-column-tile GEMM (mfma_nodes32 MODE 2) with the weight fragments read from LDS instead of held in
// registers: wl = this wave's block [NKS/4 quads][64 lanes][4] (lane-linear 16-byte reads, conflict-free), filled once per launch.
// Role B of the SPEC forward kernel needs its candidate weights for nothing else, and 48 registers for the spectral mixes.
template <int NKS>
__device__ __forceinline__ float mfma_rem4_ldsw(const float* __restrict__ X, int stride, int lane, int lg, const float* __restrict__ wl) {
    const int s1 = lg ^ sigma4(lane & 3);
    const float* p1 = X + (16 + (lane & 3)) * stride;
    auto frag = [&](int q) { return *reinterpret_cast<const float4*>(p1 + 64 * (q >> 2) + 4 * ((4 * (q & 3)) ^ s1)); };
    f32x4 rem[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rem[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NQ = NKS / 4, D = 2;                       // fragments D quads ahead: a quad is only 4 x 8 cycles of MFMAs
    float4 xa[D], wa[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        xa[d] = frag(d < NQ ? d : NQ - 1);
        wa[d] = *reinterpret_cast<const float4*>(wl + (d < NQ ? d : NQ - 1) * 256 + 4 * lane);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 a1 = xa[q % D], w4 = wa[q % D];
        if (q + D < NQ) {
            xa[q % D] = frag(q + D);
            wa[q % D] = *reinterpret_cast<const float4*>(wl + (q + D) * 256 + 4 * lane);
        }
        EEG_SCHED_FENCE();
        const float x1[4] = {a1.x, a1.y, a1.z, a1.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) rem[j] = mfma4(x1[j], ww[j], rem[j]);
        EEG_SCHED_FENCE();
    }
    f32x4 t = (rem[0] + rem[1]) + (rem[2] + rem[3]);
    EEG_PIN(t);
    return rem4_reduce(t);
}

// Optional in-kernel phase timer (development aid, eeg_dcrnn_set_seq_probe): lane 0 of every wave
// accumulates shader-clock cycles per phase.  COMPILE-TIME switch: a run-time "probe != nullptr"
// branch directly behind an MFMA chain lets the compiler sink the first VALU read of the MFMA
// result below the branch, where its hazard recognizer no longer pads the XDL-write -> VALU-read
// wait states (observed on gfx950 / ROCm 7.2: components 2,3 of the accumulator read stale).
// Clock sample of a product launch (eeg_dcrnn_prof_clock_samples): ONE lane of the chip reads the shader-clock and the 100 MHz
// real-time counters in front of and behind its time loop and adds the differences to clk[0] / clk[1] -- the clock the part holds
// INSIDE this kernel (2.37-2.39 GHz in steady state, 1.9-2.2 GHz while a fresh process ramps up).  Outside the step loop.
struct ClockSample {
    long long c0, r0;
    __device__ __forceinline__ void begin(const long long* clk, bool me) {
        if (clk != nullptr && me) { c0 = cycle_now(); r0 = realtime_now(); }
    }
    __device__ __forceinline__ void end(long long* clk, bool me) {
        if (clk != nullptr && me) {
            atomic_add_u64(reinterpret_cast<unsigned long long*>(clk), (unsigned long long)(cycle_now() - c0));
            atomic_add_u64(reinterpret_cast<unsigned long long*>(clk) + 1, (unsigned long long)(realtime_now() - r0));
        }
    }
};

template <bool ON>
struct PhaseProbe {
    long long acc[8];
    long long last;
    long long rt[4];            // chip-wide 100 MHz counter at kernel entry / first step / behind the last step / kernel exit
    __device__ __forceinline__ void start() {
        if (ON) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0;
            rt[0] = realtime_now();
            rt[1] = rt[2] = rt[3] = 0;
            last = cycle_now();
        }
    }
    __device__ __forceinline__ void mark(int k) {
        if (ON) {
            const long long t = cycle_now();
            acc[k] += t - last;
            last = t;
        }
    }
    __device__ __forceinline__ void stamp(int k) {
        if (ON) rt[k] = realtime_now();
    }
    __device__ __forceinline__ void dump(long long* p, int slot0) {
        if (ON) {
            rt[3] = realtime_now();
            if (p != nullptr && (threadIdx.x & 63) == 0) {
                long long* d = p + ((size_t)blockIdx.x * 4 + ((threadIdx.x >> 6) & 3)) * 32 + slot0;
#pragma unroll
                for (int i = 0; i < 8; ++i) d[i] = acc[i];
                long long* e = p + ((size_t)blockIdx.x * 4 + ((threadIdx.x >> 6) & 3)) * 32 + 16 + slot0 / 2;     // 16..19 forward, 20..23 backward
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = rt[i];
            }
        }
    }
};

template <int H, int M, int NKS, bool PROBE = false>
__global__ __launch_bounds__(256, 1) void seq_fwd_kernel(
    const float* __restrict__ XW, const float* __restrict__ h0, const float* __restrict__ P, int p_batched,
    const float* __restrict__ bhg, const float* __restrict__ bhc,
    float* __restrict__ Hseq, float* __restrict__ Rs, float* __restrict__ Us, float* __restrict__ Cs,
    float* __restrict__ RHs, float* __restrict__ Hpl, float* __restrict__ RHpl, size_t plane_stride,
    int T, int B, int N, int act, long long* probe) {
    using G = SeqGeom<H, M>;
    constexpr int KAP = G::KAP, KS = G::KS, CT = G::CT, NGT = G::NGT, NCT = G::NCT;
    PhaseProbe<PROBE> pp;
    pp.start();
    EEG_DYN_SMEM(sm);
    float* Pl = sm;
    float* A = Pl + (M - 1) * kPFloats;     // [32][KAP]  slot 0 = h, slots m = P_m h
    float* A2 = A + 32 * KAP;               // [32][KAP]  slot 0 = r*h
    constexpr bool REM4 = NKS == 5;         // at most 20 nodes: the second node tile runs as 4x4x1 MFMAs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const bool save = Rs != nullptr;

    // Wave w owns column tiles ct = w + 4*i of r, u, c and h (so gate tiles ct and NCT+ct): the
    // epilogue -> diffusion hand-offs are wave-local and u never leaves registers.
    // recurrent weights -> registers (MFMA fragments), once for all T steps
    float wg[2 * CT][KS], wc[CT][KS];       // gate fragments: [i] = r tile, [CT + i] = u tile
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int ct = wave + 4 * i < NCT ? wave + 4 * i : 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wg[i][ks] = bhg[((size_t)ks * NGT + ct) * 64 + lane];
            wg[CT + i][ks] = bhg[((size_t)ks * NGT + NCT + ct) * 64 + lane];
            wc[i][ks] = bhc[((size_t)ks * NCT + ct) * 64 + lane];
        }
    }

    // A workgroup walks clips b = blockIdx.x, + gridDim.x, ... (the host launches min(B, #CUs) workgroups): the
    // register-resident weights are fetched once per workgroup, not once per clip -- what matters for the
    // decoder's single-step launches at batches beyond one clip per CU.
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();                                                // previous clip: all waves done with A / A2 / Pl
    for (int e = tid; e < 2 * 32 * KAP; e += 256) A[e] = 0.f;
    lds_load_polys(Pl, P, p_batched ? b : 0, M, N);
    __syncthreads();
    float pf[poly_slots<M, NKS>()][NKS];
    load_poly_frags<M, NKS, false>(Pl, pf, lr, lg);
    if (h0 != nullptr) {
        for (int e = tid; e < N * H; e += 256) A[lds_sw(e / H, e % H, KAP)] = h0[(size_t)b * N * H + e];
    } else {      // zero initial state: clear this clip's row of the slot in front of Hseq (= Hext slot 0, read by the backward)
        for (int e = tid; e < N * H; e += 256) (Hseq - (size_t)B * N * H)[(size_t)b * N * H + e] = 0.f;
    }
    __syncthreads();

    // nodes owned by this lane (per node tile), clamped copies for branch-free loads
    const int node[2] = {lr, 16 + lr};
    const bool valid[2] = {lr < N, 16 + lr < N};
    const int nodec[2] = {valid[0] ? lr : N - 1, valid[1] ? 16 + lr : N - 1};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int oxw[CT][2], oh[CT][2];              // 32-bit element offsets inside one time step
    // REM4: the remainder nodes 16..19 are handled one element per lane (mfma_nodes32 L1): lane (lr, lg) <-> node 16 + lg,
    // column ct*16 + lr; oxw[.][1] / oh[.][1] / l1[.] are then that element's offsets
    const int node1 = 16 + lg;
    const bool valid1 = node1 < N;
    int l1[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int ct = wave + 4 * i < NCT ? wave + 4 * i : 0;
            oxw[i][nt] = nodec[nt] * (3 * H) + ct * 16 + 4 * lg;
            oh[i][nt] = nodec[nt] * H + ct * 16 + 4 * lg;
            if (REM4 && nt == 1) {
                oxw[i][1] = (valid1 ? node1 : N - 1) * (3 * H) + ct * 16 + lr;
                oh[i][1] = (valid1 ? node1 : N - 1) * H + ct * 16 + lr;
                l1[i] = lds_sw(node1, ct * 16 + lr, KAP);
            }
        }

    // hop-diffuse this wave's own column tiles of buf; planes != nullptr: the hop rows of step t also go to
    // global memory (Hpl / RHpl, the A operands of the hoisted weight-gradient GEMMs)
    const bool plane_buf = (double)(M - 1) * plane_stride * sizeof(float) < 2147483648.0;   // a 2 GB descriptor reaches every hop plane
    auto diffuse_own = [&](float* buf, float* planes, int t) {
        EEG_WAVE_SYNC();
        float* g = planes != nullptr ? planes + ((size_t)t * B + b) * N * H : nullptr;
#pragma unroll
        for (int i = 0; i < CT; ++i)
            if (wave + 4 * i < NCT) lds_diffuse_tile<M, NKS>(buf, KAP, (wave + 4 * i) * 16, H, pf, lr, lg, g, plane_stride, N, plane_buf);
    };
    diffuse_own(A, Hpl, 0);                                         // hops(h_0)
    // The hoisted pre-activations of step t+1 are fetched in the middle of step t, AHEAD of that step's
    // h / c stores in the memory queue, so that waiting for them does not have to drain those stores.
    f32x4 nxr[CT][2], nxu[CT][2], nxc[CT][2];
    // operands and results of a step go through buffer descriptors on the step's rows (scalar base, one 32-bit lane offset per
    // tile): 64-bit per-lane addresses cost two VALU instructions per access, and VALU time is matrix-pipe time here
    auto fetch_xw = [&](int t) {
        const wbuf_t bx = make_wbuf(XW + ((size_t)t * B + b) * N * (3 * H));
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (REM4 && nt == 1) {             // one element per lane: component 0
                    nxr[i][1] = (f32x4){wbuf_ld(bx, oxw[i][1], 0u), 0.f, 0.f, 0.f};
                    nxu[i][1] = (f32x4){wbuf_ld(bx, oxw[i][1] + H, 0u), 0.f, 0.f, 0.f};
                    nxc[i][1] = (f32x4){wbuf_ld(bx, oxw[i][1] + 2 * H, 0u), 0.f, 0.f, 0.f};
                    continue;
                }
                nxr[i][nt] = wbuf_ld4(bx, oxw[i][nt], 0u);
                nxu[i][nt] = wbuf_ld4(bx, oxw[i][nt] + H, 0u);
                nxc[i][nt] = wbuf_ld4(bx, oxw[i][nt] + 2 * H, 0u);
            }
    };
    fetch_xw(0);
    for (int t = 0; t < T; ++t) {
        const size_t s = (size_t)t * B + b;
        f32x4 ag[2 * CT][2], ac[CT][2], ug[CT][2];
        // the accumulators start from the hoisted pre-activations (no zero fill, no add behind the GEMM); with the 4x4x1
        // remainder the second tile's accumulator is one value per lane (component 0)
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                ag[i][nt] = nxr[i][nt]; ag[CT + i][nt] = nxu[i][nt]; ac[i][nt] = nxc[i][nt];
            }
        EEG_LDS_BARRIER();                                            // (1) hops(h) complete
        pp.mark(0);

        // gate GEMM: (2H cols) x (32 nodes), K = M*H
        mfma_nodes32<2 * CT, KS, REM4, 0, 0, false, false, REM4>(A, KAP, lane, lr, lg, wg, ag);
        pp.mark(1);
        const wbuf_t bR = make_wbuf((save ? Rs : Hseq) + s * N * H), bRH = make_wbuf((save ? RHs : Hseq) + s * N * H),
                     bU = make_wbuf((save ? Us : Hseq) + s * N * H);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ct = wave + 4 * i;
            if (ct < NCT) {                                          // wave-uniform
                const int col = ct * 16 + 4 * lg;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (REM4 && nt == 1) {                           // node 16 + lg, column ct*16 + lr
                        const float rg1 = sigmoidf_(ag[i][1][0]), u1 = sigmoidf_(ag[CT + i][1][0]);
                        ug[i][1] = (f32x4){u1, 0.f, 0.f, 0.f};
                        const float rh1 = valid1 ? rg1 * A[l1[i]] : 0.f;
                        A2[l1[i]] = rh1;
                        if (save && valid1) {
                            wbuf_st1(bR, oh[i][1], 0u, rg1);
                            wbuf_st1(bRH, oh[i][1], 0u, rh1);
                            wbuf_st1(bU, oh[i][1], 0u, u1);
                        }
                        continue;
                    }
                    const f32x4 rg = sigmoid4_(ag[i][nt]), u = sigmoid4_(ag[CT + i][nt]);
                    ug[i][nt] = u;
                    f32x4 rh = rg * ld4(A + lds_sw(node[nt], col, KAP));
                    rh = valid[nt] ? rh : zero4;
                    st4(A2 + lds_sw(node[nt], col, KAP), rh);
                    if (save && valid[nt]) {
                        wbuf_st4(bR, oh[i][nt], 0u, rg);
                        wbuf_st4(bRH, oh[i][nt], 0u, rh);
                        wbuf_st4(bU, oh[i][nt], 0u, u);
                    }
                }
            }
        }
        pp.mark(2);
        diffuse_own(A2, RHpl, t);                                   // own column tiles: no barrier needed
        pp.mark(7);
        EEG_LDS_BARRIER();                                            // (2) hops(r*h) complete
        pp.mark(3);
        if (t + 1 < T) fetch_xw(t + 1);

        // candidate GEMM: (H cols) x (32 nodes), K = M*H
        mfma_nodes32<CT, KS, REM4, 0, 0, false, false, REM4>(A2, KAP, lane, lr, lg, wc, ac);
        pp.mark(4);
        const wbuf_t bH = make_wbuf(Hseq + s * N * H), bC = make_wbuf((save ? Cs : Hseq) + s * N * H);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ct = wave + 4 * i;
            if (ct < NCT) {
                const int col = ct * 16 + 4 * lg;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (REM4 && nt == 1) {
                        const float u1 = ug[i][1][0], h1 = A[l1[i]];
                        const float c1 = act == 0 ? tanhf_(ac[i][1][0]) : fmaxf(ac[i][1][0], 0.f);
                        const float hn1 = valid1 ? u1 * h1 + (1.f - u1) * c1 : 0.f;
                        A[l1[i]] = hn1;
                        if (valid1) {
                            wbuf_st1(bH, oh[i][1], 0u, hn1);
                            if (save) wbuf_st1(bC, oh[i][1], 0u, c1);
                        }
                        continue;
                    }
                    const f32x4 u = ug[i][nt], h = ld4(A + lds_sw(node[nt], col, KAP));
                    const f32x4 c = act == 0 ? tanh4_(ac[i][nt]) : relu4_(ac[i][nt]);
                    f32x4 hn = u * h + (1.f - u) * c;
                    hn = valid[nt] ? hn : zero4;
                    st4(A + lds_sw(node[nt], col, KAP), hn);
                    if (valid[nt]) {
                        wbuf_st4(bH, oh[i][nt], 0u, hn);
                        if (save) wbuf_st4(bC, oh[i][nt], 0u, c);
                    }
                }
            }
        }
        pp.mark(5);
        // hops(h_t) of the own column tiles for the next step (their slot-0 source was just written by
        // this wave; other waves only read A2 until barrier (1) of the next step)
        if (t + 1 < T || Hpl != nullptr) diffuse_own(A, Hpl, t + 1);   // slot T = hops(h_{T-1}): the next layer's input planes
        pp.mark(6);
    }
    }   // clips of this workgroup
    pp.dump(probe, 0);
}

// ---- two waves per SIMD (rnn_units = 64, at most 20 nodes) -----------------------------------------------------
// Same step as seq_fwd_kernel, but 8 waves: wave w (0..3, role A) owns column tile w of r, r*h and the 16-node
// part of c and h; wave 4+w (role B, same SIMD) owns column tile w of the UPDATE gate u -- which depends on nothing
// but hops(h) and is only needed by the final blend -- and the remainder nodes 16..19 of c and h.  The matrix pipe
// of a SIMD is shared, so this adds no MFMA capacity: it takes work off the critical chain (r -> r*h -> c -> h) and
// lets one wave's LDS reads, hand-overs and epilogue latencies be filled by the other wave's MFMAs.
//   phase 1 (after barrier 1: hops(h) complete)   A: r (both node tiles), r*h, hops(r*h)      B: u (nodes 0..15 -> LDS U)
//   phase 2 (after barrier 2: hops(r*h) complete) A: c, h' of nodes 0..15 (u from LDS U)      B: c, h' of nodes 16..19
//   phase 3 (after barrier 3: h' complete)        A: hops(h') for the next step              B: -
// SPEC (spectral form of the hoisted x-part, spec_common.h): the pre-activations arrive in the eigenbasis of the shared support,
// Yh (N, spec_Sp, 3H) node-major with the bias already inside (row of (t, b) = t*B + b), and role B turns its column tile of every
// step into [r|u|c] = U Yh itself: 15 dword loads a step ahead, 15 + 15 MFMAs behind barrier (3); r and the 16-node part of c go to
// role A through two [16][64] LDS tiles (XR, XC; the remainder of r through XRr), u and the remainder of c stay in its registers.
// The by-products the backward wants leave in the eigenbasis too, from role B: Hh (N, spec_SpE, H) row slot*B + b <- U^T h_slot
// (slot 0 = the initial state; rows B.. of every frequency ARE the next layer's transformed input) and RHh (N, spec_Sp, H) <-
// U^T (r * h_{t-1}) -- 10 + 10 MFMAs per step instead of the hop planes.  Three HBM passes (U Yh, U^T h, U^T (r*h)) are gone.
template <int H, int M, int NKS, bool PROBE = false, bool SPEC = false>
__global__ __launch_bounds__(512, 1) void seq_fwd2_kernel(
    const float* __restrict__ XW, const float* __restrict__ h0, const float* __restrict__ P, int p_batched,
    const float* __restrict__ bhg, const float* __restrict__ bhc,
    float* __restrict__ Hseq, float* __restrict__ Rs, float* __restrict__ Us, float* __restrict__ Cs,
    float* __restrict__ RHs, float* __restrict__ Hpl, float* __restrict__ RHpl, size_t plane_stride,
    int T, int B, int N, int act, long long* probe, const float* __restrict__ spec_U = nullptr, int spec_Sp = 0, int spec_SpE = 0) {
    using G = SeqGeom<H, M>;
    PhaseProbe<PROBE> pp;
    pp.start();
    static_assert(G::CT == 1 && NKS == 5, "one column tile per wave, second node tile on the 4x4x1 MFMA");
    constexpr int KAP = G::KAP, KS = G::KS, NGT = G::NGT, NCT = G::NCT, UST = 64;    // U: swizzled like the big tiles
    EEG_DYN_SMEM(sm);
    float* Pl = sm;
    float* A = Pl + (M - 1) * kPFloats;     // [32][KAP]  slot 0 = h, slots m = P_m h
    float* A2 = A + 32 * KAP;               // [32][KAP]  slot 0 = r*h
    // (SPEC: the role is a scalar -- with a per-lane role the two role bodies are linearised one behind the other and what one
    //  of them keeps in registers for the whole launch stays allocated while the other runs)
    const int tid = threadIdx.x, lane = tid & 63, wave8 = SPEC ? wave_uniform(tid >> 6) : (tid >> 6), role = wave8 >> 2, wave = wave8 & 3;
    const int lr = lane & 15, lg = lane >> 4;
    float* U = A2 + 32 * KAP;                      // [16][UST] update gate of nodes 0..15 of the current step
    float* XR = U + 16 * UST;                      // SPEC: [16][UST] r pre-activations of nodes 0..15 of the next step (from role B)
    float* XC = XR + 16 * UST;                     //       ... of c, two buffers (step parity: written one phase after the other is read)
    float* XRr = XC + 2 * 16 * UST;                //       [4 tiles][64 lanes]: r pre-activation of node 16 + lg, column ct*16 + lr
    float* W1L = XRr + 4 * 64;                     //       [4 tiles][KS/4][64][4]: role B's candidate weights (mfma_rem4_ldsw)
    const bool save = Rs != nullptr;
    const int ct = wave;                               // NCT == 4 == waves per role
    // results leave through buffer descriptors (one VGPR offset per node tile, the step offset in an SGPR): the 64-bit
    // per-lane address arithmetic of ~10 stores per step is VALU work, and VALU shares its ALUs with the fp32 MFMAs
    const wbuf_t bH = make_wbuf(Hseq), bR = make_wbuf(save ? Rs : Hseq), bU = make_wbuf(save ? Us : Hseq),
                 bC = make_wbuf(save ? Cs : Hseq), bRH = make_wbuf(save ? RHs : Hseq);
    // (the launcher takes this kernel only while T*B*N*H floats stay below 2 GB: 32-bit buffer offsets)

    float w0[1][KS], w1[1][KS];                        // role A: r and c fragments; role B: u and c fragments (SPEC: see below)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) w0[0][ks] = bhg[((size_t)ks * NGT + (role == 0 ? ct : NCT + ct)) * 64 + lane];
    if (!SPEC || role == 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w1[0][ks] = bhc[((size_t)ks * NCT + ct) * 64 + lane];
    } else {                                           // SPEC, role B: its candidate fragments go to LDS once (quad-major, lane-linear)
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = bhc[((size_t)(4 * q + j) * NCT + ct) * 64 + lane];
            st4(W1L + (ct * (KS / 4) + q) * 256 + 4 * lane, v);
        }
    }
    // One clip of this workgroup.  ROLE_C >= 0 (SPEC): the role is a compile-time constant of the call and each role walks the clips
    // in a loop of its own -- inside ONE loop the other role's body is reachable from every point of a role's body through the loop
    // header, so everything that role keeps in registers for the whole launch (role A's candidate fragments) would stay allocated here.
    auto clip = [&](const int b, auto ROLE_T) __attribute__((always_inline)) {
    constexpr int ROLE_C = decltype(ROLE_T)::value;
    __syncthreads();
    for (int e = tid; e < 2 * 32 * KAP; e += 512) A[e] = 0.f;
    lds_load_polys(Pl, P, p_batched ? b : 0, M, N);
    __syncthreads();
    if (h0 != nullptr) {
        for (int e = tid; e < N * H; e += 512) A[lds_sw(e / H, e % H, KAP)] = h0[(size_t)b * N * H + e];
    } else {
        for (int e = tid; e < N * H; e += 512) (Hseq - (size_t)B * N * H)[(size_t)b * N * H + e] = 0.f;
    }
    __syncthreads();

    const bool valid[2] = {lr < N, 16 + lr < N};
    const int nodec[2] = {valid[0] ? lr : N - 1, valid[1] ? 16 + lr : N - 1};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int col = ct * 16 + 4 * lg;
    const int oxw0 = nodec[0] * (3 * H) + col, oh0 = nodec[0] * H + col;
    // remainder nodes 16..19, one value per lane (mfma_nodes32 L1): lane (lr, lg) <-> node 16 + lg, column ct*16 + lr
    const int node1 = 16 + lg, col1 = ct * 16 + lr;
    const bool valid1 = node1 < N;
    const int oxw1 = (valid1 ? node1 : N - 1) * (3 * H) + col1, oh1 = (valid1 ? node1 : N - 1) * H + col1;
    const int l1 = lds_sw(node1, col1, KAP);           // this lane's element of the h / r*h slot (rows 16..19)
    if (ROLE_C < 0 ? role == 0 : ROLE_C == 0) {
        EEG_SETPRIO(3);         // the r -> r*h -> c chain is the critical path: its instructions issue first
        float pf[poly_slots<M, NKS>()][NKS];
        load_poly_frags<M, NKS, false>(Pl, pf, lr, lg);
        auto diffuse_own = [&](float* buf, float* planes, int t) {
            EEG_WAVE_SYNC();
            float* g = (!SPEC && planes != nullptr) ? planes + ((size_t)t * B + b) * N * H : nullptr;   // (SPEC: Hpl / RHpl are role B's U^T h, U^T (r*h))
            lds_diffuse_tile<M, NKS>(buf, KAP, ct * 16, H, pf, lr, lg, g, plane_stride, N, true);
        };
        diffuse_own(A, Hpl, 0);
        f32x4 nxr0 = zero4, nxc = zero4;
        float nxr1 = 0.f;
        auto fetch_xw = [&](int t) {                                  // (descriptor on the step's rows: no 64-bit lane addresses)
            if constexpr (SPEC) return;                               // (role B mixes them out of Yh: XR / XC / XRr)
            const wbuf_t bx = make_wbuf(XW + ((size_t)t * B + b) * N * (3 * H));
            nxr0 = wbuf_ld4(bx, oxw0, 0u);
            nxr1 = wbuf_ld(bx, oxw1, 0u);
            nxc = wbuf_ld4(bx, oxw0 + 2 * H, 0u);
        };
        fetch_xw(0);
        pp.stamp(1);
        pp.last = PROBE ? cycle_now() : 0;
        ClockSample cs;
        const bool cs_me = !PROBE && b == 0 && tid == 0;         // (the probe instantiation uses `probe` for its own records)
        cs.begin(probe, cs_me);
        // Like role B's update gate (below), the hop-0 slot of the NEXT step's r GEMM runs ahead, behind this wave's node mix in the
        // third window (its LDS writes drain meanwhile); the two-pass stream of the other slots picks the chains up (QB / carry).
        constexpr int QSA = KS / 4 / M;
        f32x4 rc[5] = {zero4, zero4, zero4, zero4, zero4};
        f32x4 ra = zero4;
        bool pre_r = false;
        for (int t = 0; t < T; ++t) {
            const size_t s = (size_t)t * B + b;
            // the accumulators start from the hoisted pre-activations (no zero fill, no add behind the GEMM); the
            // remainder's is one value per lane (component 0)
            f32x4 ar[1][2] = {{nxr0, (f32x4){nxr1, 0.f, 0.f, 0.f}}}, ac[1][2] = {{nxc, zero4}};
            const unsigned so = (unsigned)(s * N * H);
            EEG_LDS_BARRIER();                                        // (1) hops(h) complete
            pp.mark(0);
            if constexpr (SPEC) {                                     // this step's r pre-activations, mixed by role B in the window before
                ar[0][0] = ld4(XR + lds_sw(lr, col, UST));
                ar[0][1][0] = XRr[ct * 64 + lane];
                if (M > 1 && pre_r) ra += ar[0][0];                   // (the run-ahead hop-0 slot started from zero)
            }

            // all 16x16x4 MFMAs of the r GEMM, then all its 4x4x1 MFMAs (TWOPASS: one change of MFMA shape instead of one per quad;
            // round 4, with the register-reduced remainder: seq_fwd 0.527 -> 0.512 ms; the same order for role B's u GEMM loses)
            if (M > 1 && pre_r) {
                ar[0][0] = ra;
                mfma_nodes32<1, KS, true, 0, 0, false, true, true, (M > 1 ? QSA : 0)>(A, KAP, lane, lr, lg, w0, ar, rc);
            } else
                mfma_nodes32<1, KS, true, 0, 0, false, true, true>(A, KAP, lane, lr, lg, w0, ar);
            pp.mark(1);
            {
                const f32x4 rg = sigmoid4_(ar[0][0]);
                f32x4 rh = rg * ld4(A + lds_sw(lr, col, KAP));
                rh = valid[0] ? rh : zero4;
                st4(A2 + lds_sw(lr, col, KAP), rh);
                const float rg1 = sigmoidf_(ar[0][1][0]);
                const float rh1 = valid1 ? rg1 * A[l1] : 0.f;
                A2[l1] = rh1;
                if (save && valid[0]) {
                    wbuf_st4(bR, oh0, so, rg);
                    wbuf_st4(bRH, oh0, so, rh);
                }
                if (save && valid1) {
                    wbuf_st1(bR, oh1, so, rg1);
                    wbuf_st1(bRH, oh1, so, rh1);
                }
            }
            pp.mark(2);
            diffuse_own(A2, RHpl, t);
            pp.mark(7);
            EEG_LDS_BARRIER();                                        // (2) hops(r*h) and u of nodes 0..15 complete
            pp.mark(3);
            if (t + 1 < T) fetch_xw(t + 1);
            if constexpr (SPEC) ac[0][0] = ld4(XC + (t & 1) * (16 * UST) + lds_sw(lr, col, UST));
            mfma_nodes32<1, KS, true, 1>(A2, KAP, lane, lr, lg, w1, ac);
            pp.mark(4);
            {
                const f32x4 u = ld4(U + lds_sw(lr, col, UST)), h = ld4(A + lds_sw(lr, col, KAP));
                const f32x4 c = act == 0 ? tanh4_(ac[0][0]) : relu4_(ac[0][0]);
                f32x4 hn = u * h + (1.f - u) * c;
                hn = valid[0] ? hn : zero4;
                st4(A + lds_sw(lr, col, KAP), hn);
                if (valid[0]) {
                    wbuf_st4(bH, oh0, so, hn);
                    if (save) wbuf_st4(bC, oh0, so, c);
                }
            }
            pp.mark(5);
            EEG_LDS_BARRIER();                                        // (3) h' complete (rows 16..19 come from role B)
            if (t + 1 < T || (!SPEC && Hpl != nullptr)) diffuse_own(A, Hpl, t + 1);
            pre_r = M > 1 && t + 1 < T;
            if (pre_r) {      // hop-0 slot of the next step's r GEMM (seq_fwd -2.8 %, profiles/r04_d_*)
                ra = SPEC ? zero4 : nxr0;
                rc[0] = rc[1] = rc[2] = rc[3] = rc[4] = zero4;
                f32x4 rr4[4] = {zero4, zero4, zero4, zero4};
                mfma_tile_quads<KS, 0, QSA>(A, KAP, lane, lr, lg, w0[0], ra, rc[4], rr4);
                rc[0] = rr4[0]; rc[1] = rr4[1]; rc[2] = rr4[2]; rc[3] = rr4[3];
            }
            pp.mark(6);
        }
        cs.end(probe, cs_me);
        pp.stamp(2);
    } else {
        f32x4 nxu0 = zero4;
        float nxu1 = 0.f, nxc1 = 0.f;
        // SPEC: this lane's fragments of U (pre-activations: rows of U as the MFMA B operand) and of U^T (by-products), the Yh values of
        // the next step (node rows 4ks + lg of this column tile of r, u, c), and the two mixes
        float ux[NKS], ux4[NKS], ut[NKS], ut4[NKS], yv[3][NKS];
        // (16 <= N <= 20 here: node rows 4ks + lg of the first four k-steps exist for every lane -- one lane offset + a scalar per k-step;
        //  the rows 16 + lg of the last one are clamped to N - 1 where they do not exist: ux / ux4 are zero there)
        const unsigned vyb = (unsigned)lg * (unsigned)spec_Sp * (3 * H) + ct * 16 + lr;
        const unsigned vy4 = (unsigned)(16 + lg < N ? 16 + lg : N - 1) * (unsigned)spec_Sp * (3 * H) + ct * 16 + lr;
        const wbuf_t bY = make_wbuf(XW), bHh = make_wbuf(Hpl != nullptr ? Hpl : Hseq), bRHh = make_wbuf(RHpl != nullptr ? RHpl : Hseq);
        const unsigned oe0 = (unsigned)lr * (unsigned)spec_SpE * H + col, oe1 = (unsigned)node1 * (unsigned)spec_SpE * H + col1;
        const unsigned or0 = (unsigned)lr * (unsigned)spec_Sp * H + col, or1 = (unsigned)node1 * (unsigned)spec_Sp * H + col1;
        if constexpr (SPEC) {
            load_spec_frags<NKS>(spec_U, N, lane, ut, ut4);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int i = 4 * ks + lg, j4 = 16 + (lane & 3);
                ux[ks] = (i < N && lr < N) ? spec_U[lr * N + i] : 0.f;
                ux4[ks] = (i < N && j4 < N) ? spec_U[j4 * N + i] : 0.f;
            }
        }
        auto fetch_y = [&](int t) {
            const unsigned so = (unsigned)(((size_t)t * B + b) * (3 * H));
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    yv[g][ks] = ks < 4 ? wbuf_ld(bY, vyb + g * H, so + (unsigned)(4 * ks) * (unsigned)spec_Sp * (3 * H)) : wbuf_ld(bY, vy4 + g * H, so);
        };
        auto xw_mix = [&](int tn) {                                  // yv (step tn) -> XR, XC[tn & 1], XRr (LDS, for role A) and nxu0, nxu1, nxc1
            f32x4 acc[3] = {zero4, zero4, zero4}, rem[3] = {zero4, zero4, zero4}, rem2[3] = {zero4, zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = mfma16(yv[g][ks], ux[ks], acc[g]);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    if (ks & 1) rem2[g] = mfma4(ux4[ks], yv[g][ks], rem2[g]);
                    else rem[g] = mfma4(ux4[ks], yv[g][ks], rem[g]);
                }
            // (rem4_reduce's register swaps are inline asm: never the first readers of an MFMA result -- see spec_mix_tiles_out)
            f32x4 t0 = rem[0] + rem2[0], t1 = rem[1] + rem2[1], t2 = rem[2] + rem2[2];
            EEG_PIN(t0); EEG_PIN(t1); EEG_PIN(t2);
            st4(XR + lds_sw(lr, col, UST), acc[0]);
            st4(XC + (tn & 1) * (16 * UST) + lds_sw(lr, col, UST), acc[2]);
            XRr[ct * 64 + lane] = rem4_reduce(t0);
            nxu0 = acc[1];
            nxu1 = rem4_reduce(t1);
            nxc1 = rem4_reduce(t2);
        };
        auto export_tile = [&](const float* tile, wbuf_t out, unsigned o0, unsigned o1, unsigned row) {
            const float* const tl[1] = {tile};
            const int strd[1] = {KAP}, scol[1] = {ct * 16}, gcol[1] = {0};
            spec_mix_tiles_out<NKS, 1>(tl, strd, scol, gcol, ut, ut4, lr, lg, out, o0, o1, row * H, valid[0], valid1);
        };
        auto fetch_x = [&](int t) {
            if constexpr (SPEC) { fetch_y(t); return; }
            const wbuf_t bx = make_wbuf(XW + ((size_t)t * B + b) * N * (3 * H));
            nxu0 = wbuf_ld4(bx, oxw0 + H, 0u);
            nxu1 = wbuf_ld(bx, oxw1 + H, 0u);
            nxc1 = wbuf_ld(bx, oxw1 + 2 * H, 0u);
        };
        fetch_x(0);
        if constexpr (SPEC) {
            xw_mix(0);                                                // the pre-activations of step 0
        }
        // The hop-0 slot of the NEXT step's update-gate GEMM (the first K quads: h' itself, complete at barrier 3) runs in the third
        // window, where role A mixes h' and the matrix pipe is otherwise idle; the other hop slots follow behind barrier (1).
        // (Round 3 measured this move as a loss; with the register-reduced remainder it wins: seq_fwd -3..4 %, profiles/r04_d_*.)
        constexpr int QS = KS / 4 / M;                               // quads of one hop slot
        f32x4 ua = zero4, ub = zero4, urem[4] = {zero4, zero4, zero4, zero4};
        float ux1 = 0.f;
        bool pre = false;
        for (int t = 0; t < T; ++t) {
            const size_t s = (size_t)t * B + b;
            f32x4 au[1][2] = {{nxu0, (f32x4){nxu1, 0.f, 0.f, 0.f}}}, ac[1][2] = {{zero4, (f32x4){nxc1, 0.f, 0.f, 0.f}}};
            const unsigned so = (unsigned)(s * N * H);
            EEG_LDS_BARRIER();                                        // (1)
            if (!pre) {                                               // first step: nothing was run ahead
                ua = nxu0; ub = zero4; ux1 = nxu1;
                urem[0] = urem[1] = urem[2] = urem[3] = zero4;
            }
            if (t + 1 < T) fetch_x(t + 1);
            if (!pre) mfma_tile_quads<KS, 0, QS>(A, KAP, lane, lr, lg, w0[0], ua, ub, urem);
            mfma_tile_quads<KS, QS, KS / 4>(A, KAP, lane, lr, lg, w0[0], ua, ub, urem);
            {
                f32x4 tt = (urem[0] + urem[1]) + (urem[2] + urem[3]);
                EEG_PIN(tt);
                au[0][0] = ua + ub;
                au[0][1][0] = ux1 + rem4_reduce(tt);
            }
            float u1;                                                // node 16 + lg: stays in a register for the blend
            {
                const f32x4 u0 = sigmoid4_(au[0][0]);
                u1 = sigmoidf_(au[0][1][0]);
                st4(U + lds_sw(lr, col, UST), u0);                        // nodes >= N: finite, never used
                if (save && valid[0]) wbuf_st4(bU, oh0, so, u0);
            }
            if constexpr (SPEC) {                                     // slot t of Hh: U^T h_{t-1} (slot 0 of A: complete since barrier 3, rewritten in phase 2)
                if (Hpl != nullptr) export_tile(A, bHh, oe0, oe1, (unsigned)s);
            }
            EEG_LDS_BARRIER();                                        // (2)
            // remainder nodes 16..19 of this column tile: c (from hops(r*h)) and the blend, one element per lane
            if constexpr (SPEC) ac[0][1][0] += mfma_rem4_ldsw<KS>(A2, KAP, lane, lg, W1L + ct * (KS / 4) * 256);
            else mfma_nodes32<1, KS, true, 2, 0, false, false, true>(A2, KAP, lane, lr, lg, w1, ac);
            {
                const float h1 = A[l1];
                const float c1 = act == 0 ? tanhf_(ac[0][1][0]) : fmaxf(ac[0][1][0], 0.f);
                const float hn1 = valid1 ? u1 * h1 + (1.f - u1) * c1 : 0.f;
                A[l1] = hn1;                                             // rows 16..19 of this column tile (64 distinct elements)
                if (valid1) {
                    wbuf_st1(bH, oh1, so, hn1);
                    if (save) {
                        wbuf_st1(bC, oh1, so, c1);
                        wbuf_st1(bU, oh1, so, u1);
                    }
                }
            }
            if constexpr (SPEC) {
                // U^T (r * h_{t-1}) of this column tile (slot 0 of A2: complete at barrier 2), then the pre-activations of step t+1 (Yh
                // requested in phase 1): role A reads XR / XRr behind barrier (1) and the other XC buffer behind barrier (2) of step t+1
                if (RHpl != nullptr) export_tile(A2, bRHh, or0, or1, (unsigned)s);
            }
            EEG_LDS_BARRIER();                                        // (3)
#ifdef EEG_X_BPRE
            pre = t + 1 < T;
#else
            pre = !SPEC && t + 1 < T;                                 // (SPEC: the third window belongs to the mix of the next step's pre-activations)
#endif
            if constexpr (SPEC) {
                if (!pre && t + 1 < T) xw_mix(t + 1);
            }
            if (pre) {                                                // hop-0 slot of step t+1's update gate, from h' (slot 0 of A)
                ua = SPEC ? zero4 : nxu0; ub = zero4; ux1 = SPEC ? 0.f : nxu1;
                urem[0] = urem[1] = urem[2] = urem[3] = zero4;
                // (nxc1 of step t+1 stays in its register until the top of the next iteration; the registers of nxu are free now)
                mfma_tile_quads<KS, 0, QS>(A, KAP, lane, lr, lg, w0[0], ua, ub, urem);
                if constexpr (SPEC) {                                 // the pre-activations of step t+1 (their chains start from zero above)
                    xw_mix(t + 1);
                    ua += nxu0;
                    ux1 = nxu1;
                }
            }
        }
        if constexpr (SPEC) {                                         // slot T: U^T h_{T-1} (the next layer's last input row block)
            if (Hpl != nullptr) export_tile(A, bHh, oe0, oe1, (unsigned)((size_t)T * B + b));
        }
    }
    };   // clip
    if constexpr (SPEC) {
        if (role == 0) { for (int b = blockIdx.x; b < B; b += gridDim.x) clip(b, SeqIdx<0>()); }
        else { for (int b = blockIdx.x; b < B; b += gridDim.x) clip(b, SeqIdx<1>()); }
    } else {
        for (int b = blockIdx.x; b < B; b += gridDim.x) clip(b, SeqIdx<-1>());
    }
    if (role == 0) pp.dump(probe, 0);
}

// lengths: optional int64 (B); d_at_len is added at t = lengths[b]-1, d_at_end at t = T-1.
template <int H, int M, int NKS, bool PROBE = false>
__global__ __launch_bounds__(256, 1) void seq_bwd_kernel(
    const float* __restrict__ Hseq, const float* __restrict__ h0, const float* __restrict__ Rs,
    const float* __restrict__ Us, const float* __restrict__ Cs, const float* __restrict__ dHseq,
    const float* __restrict__ d_at_end, const float* __restrict__ d_at_len, const long long* __restrict__ lengths,
    const float* __restrict__ P, int p_batched, const float* __restrict__ b1p, const float* __restrict__ b2p,
    float* __restrict__ dXW, float* __restrict__ dh0, float* __restrict__ dbias_part, int T, int B, int N, int act,
    long long* probe) {
    using G = SeqGeom<H, M>;
    constexpr int KAP = G::KAP, KS = G::KS, KGP = G::KGP, KSG = G::KSG, CT = G::CT, NCT = G::NCT;
    PhaseProbe<PROBE> pp;
    pp.start();
    EEG_DYN_SMEM(sm);
    float* Pl = sm;
    constexpr int ROWS = G::bwd_rows(NKS);
    float* EC = Pl + (M - 1) * kPFloats;    // [ROWS][KAP]  slot 0 = dC, slots m = P_m^T dC
    float* EG = EC + ROWS * KAP;            // [ROWS][KGP]  slot 0 = [dR|dU], slots m = P_m^T [dR|dU]
    constexpr bool REM4 = NKS == 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;

    // wave w owns column tiles ct = w + 4*i of every H-wide quantity (and dR tile ct / dU tile ct of
    // the 2H-wide gate gradient): all elementwise -> diffusion hand-offs are wave-local.
    float w1[CT][KS], w2[CT][KSG];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int ct = wave + 4 * i < NCT ? wave + 4 * i : 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w1[i][ks] = b1p[((size_t)ks * NCT + ct) * 64 + lane];
#pragma unroll
        for (int ks = 0; ks < KSG; ++ks) w2[i][ks] = b2p[((size_t)ks * NCT + ct) * 64 + lane];
    }
    for (int b = blockIdx.x; b < B; b += gridDim.x) {               // clips of this workgroup (see seq_fwd_kernel)
    __syncthreads();                                                // previous clip: the bias reduction has read EG
    for (int e = tid; e < ROWS * (KAP + KGP); e += 256) EC[e] = 0.f;
    lds_load_polys(Pl, P, p_batched ? b : 0, M, N);
    // same clamp as gather_last_kernel: forward and backward agree on which step an out-of-range length selects
    int t_len = -1;
    if (d_at_len != nullptr) {
        t_len = lengths != nullptr ? (int)lengths[b] - 1 : T - 1;
        t_len = t_len < 0 ? 0 : (t_len >= T ? T - 1 : t_len);
    }
    __syncthreads();
    float pf[poly_slots<M, NKS>()][NKS];
    load_poly_frags<M, NKS, true>(Pl, pf, lr, lg);

    const int node[2] = {lr, 16 + lr};
    const bool valid[2] = {lr < N, 16 + lr < N};
    const int nodec[2] = {valid[0] ? lr : N - 1, valid[1] ? 16 + lr : N - 1};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int oh[CT][2], oxw[CT][2];              // 32-bit element offsets inside one time step
    bool own[CT];
    // REM4: the remainder nodes 16..19 are handled one element per lane (mfma_nodes32 L1): lane (lr, lg) <-> node 16 + lg,
    // column ct*16 + lr -- oh[.][1] / oxw[.][1] are then that element's offsets and every nt = 1 quantity below lives in
    // component 0 of its vector
    const int node1 = 16 + lg;
    const bool valid1 = node1 < N;
    int lc1[CT], lg1[CT], lu1[CT];          // LDS offsets of the element in the dC / dR / dU slots
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        own[i] = wave + 4 * i < NCT;
        const int ct = own[i] ? wave + 4 * i : 0;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            oh[i][nt] = nodec[nt] * H + ct * 16 + 4 * lg;
            oxw[i][nt] = node[nt] * (3 * H) + ct * 16 + 4 * lg;
        }
        if (REM4) {
            oh[i][1] = (valid1 ? node1 : N - 1) * H + ct * 16 + lr;
            oxw[i][1] = node1 * (3 * H) + ct * 16 + lr;
            lc1[i] = lds_sw(node1, ct * 16 + lr, KAP);
            lg1[i] = lds_sw(node1, ct * 16 + lr, KGP);
            lu1[i] = lds_sw(node1, H + ct * 16 + lr, KGP);
        }
    }

    f32x4 dh[CT][2], sb_r[CT], sb_u[CT], sb_c[CT];
    float sb1_r[CT], sb1_u[CT], sb1_c[CT];  // REM4: bias sums of the remainder element
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        dh[i][0] = zero4;
        dh[i][1] = zero4;
        sb_r[i] = sb_u[i] = sb_c[i] = zero4;
        sb1_r[i] = sb1_u[i] = sb1_c[i] = 0.f;
    }

    const size_t tstride = (size_t)B * N * H;
    const size_t boff = (size_t)b * N * H;
    // operands of step t are fetched during step t+1 (one step ahead): h_{t-1}, r, u, c and the
    // external gradient of h_t (dHseq + d_at_end + d_at_len); padding nodes read a valid row.
    f32x4 nh[CT][2], nr[CT][2], nu[CT][2], nc[CT][2], ng[CT][2];
    auto fetch = [&](int t) {
        const size_t so = (size_t)t * tstride + boff;            // wave-uniform element offset of step t
        const float* hs = t > 0 ? Hseq + (so - tstride) : (h0 != nullptr ? h0 + boff : nullptr);
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int o = oh[i][nt];
                if (REM4 && nt == 1) {
                    nh[i][1] = (f32x4){hs != nullptr ? hs[o] : 0.f, 0.f, 0.f, 0.f};
                    nr[i][1] = (f32x4){Rs[so + o], 0.f, 0.f, 0.f};
                    nu[i][1] = (f32x4){Us[so + o], 0.f, 0.f, 0.f};
                    nc[i][1] = (f32x4){Cs[so + o], 0.f, 0.f, 0.f};
                    float g1 = dHseq != nullptr ? dHseq[so + o] : 0.f;
                    if (d_at_end != nullptr && t == T - 1) g1 += d_at_end[boff + o];
                    if (t == t_len) g1 += d_at_len[boff + o];
                    ng[i][1] = (f32x4){g1, 0.f, 0.f, 0.f};
                    continue;
                }
                nh[i][nt] = hs != nullptr ? ld4(hs + o) : zero4;
                nr[i][nt] = ld4(Rs + so + o);
                nu[i][nt] = ld4(Us + so + o);
                nc[i][nt] = ld4(Cs + so + o);
                f32x4 g = dHseq != nullptr ? ld4(dHseq + so + o) : zero4;
                if (d_at_end != nullptr && t == T - 1) g += ld4(d_at_end + boff + o);
                if (t == t_len) g += ld4(d_at_len + boff + o);
                ng[i][nt] = g;
            }
    };
    fetch(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        float* dxw = dXW + ((size_t)t * B + b) * N * (3 * H);
        f32x4 hp[CT][2], rr[CT][2], dU[CT][2], dhn[CT][2], uu[CT][2], cc[CT][2], gg[CT][2];
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                hp[i][nt] = nh[i][nt]; rr[i][nt] = nr[i][nt]; uu[i][nt] = nu[i][nt]; cc[i][nt] = nc[i][nt]; gg[i][nt] = ng[i][nt];
            }
        if (t > 0) fetch(t - 1);
        pp.mark(6);
        // ---- E1: gate blend backward on the owned elements (padding nodes zeroed)
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int col = (own[i] ? wave + 4 * i : 0) * 16 + 4 * lg;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (REM4 && nt == 1) {                               // node 16 + lg, column ct*16 + lr
                    const bool ok1 = valid1 && own[i];
                    const float h1 = hp[i][1][0], u1 = uu[i][1][0], c1 = cc[i][1][0];
                    const float g1 = ok1 ? dh[i][1][0] + gg[i][1][0] : 0.f;
                    const float dc1 = g1 * (1.f - u1);
                    const float dC1 = act == 0 ? dc1 * (1.f - c1 * c1) : (c1 > 0.f ? dc1 : 0.f);
                    const float du1 = g1 * (h1 - c1) * u1 * (1.f - u1);
                    if (own[i]) EC[lc1[i]] = dC1;                    // zeros on padding nodes
                    if (ok1) {
                        dxw[oxw[i][1] + 2 * H] = dC1;
                        dxw[oxw[i][1] + H] = du1;
                    }
                    sb1_c[i] += dC1;
                    sb1_u[i] += du1;
                    dU[i][1] = (f32x4){du1, 0.f, 0.f, 0.f};
                    dhn[i][1] = (f32x4){g1 * u1, 0.f, 0.f, 0.f};
                    continue;
                }
                const bool ok = valid[nt] && own[i];
                const f32x4 h = hp[i][nt], u = uu[i][nt], c = cc[i][nt];
                const f32x4 g = ok ? dh[i][nt] + gg[i][nt] : zero4;
                f32x4 dC, du_;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dc = g[r] * (1.f - u[r]);
                    dC[r] = act == 0 ? dc * (1.f - c[r] * c[r]) : (c[r] > 0.f ? dc : 0.f);
                    du_[r] = g[r] * (h[r] - c[r]) * u[r] * (1.f - u[r]);
                }
                if (own[i] && (ROWS == 32 || node[nt] < ROWS)) st4(EC + lds_sw(node[nt], col, KAP), dC);   // zeros on padding nodes
                if (ok) {
                    st4(dxw + oxw[i][nt] + 2 * H, dC);
                    st4(dxw + oxw[i][nt] + H, du_);
                }
                sb_c[i] += dC;
                sb_u[i] += du_;
                dU[i][nt] = du_;
                dhn[i][nt] = g * u;
            }
        }
        pp.mark(0);
        EEG_WAVE_SYNC();
#pragma unroll
        for (int i = 0; i < CT; ++i)
            if (own[i]) lds_diffuse_tile<M, NKS, ROWS>(EC, KAP, (wave + 4 * i) * 16, H, pf, lr, lg);
        pp.mark(7);
        EEG_LDS_BARRIER();                                            // (1) P_m^T dC complete
        pp.mark(1);

        // ---- GEMM1: d(r*h) = [P_m^T dC]_m (32 x M*H) @ Wc^h^T (M*H x H)
        f32x4 acc[CT][2];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            acc[i][0] = zero4;
            acc[i][1] = zero4;
        }
        mfma_nodes32<CT, KS, REM4, 0, 0, false, false, REM4>(EC, KAP, lane, lr, lg, w1, acc);
        pp.mark(2);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            if (own[i]) {                                            // wave-uniform
                const int col = (wave + 4 * i) * 16 + 4 * lg;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (REM4 && nt == 1) {
                        const float drh1 = acc[i][1][0], rg1 = rr[i][1][0];       // exact 0 on padding nodes
                        const float dR1 = drh1 * hp[i][1][0] * rg1 * (1.f - rg1);
                        dhn[i][1][0] += drh1 * rg1;
                        EG[lg1[i]] = dR1;
                        EG[lu1[i]] = dU[i][1][0];
                        if (valid1) dxw[oxw[i][1]] = dR1;
                        sb1_r[i] += dR1;
                        continue;
                    }
                    const f32x4 drh = acc[i][nt], rg = rr[i][nt];    // exact 0 on padding nodes
                    const f32x4 dR = drh * hp[i][nt] * rg * (1.f - rg);
                    dhn[i][nt] += drh * rg;
                    if (ROWS == 32 || node[nt] < ROWS) {
                        st4(EG + lds_sw(node[nt], col, KGP), dR);
                        st4(EG + lds_sw(node[nt], H + col, KGP), dU[i][nt]);
                    }
                    if (valid[nt]) st4(dxw + oxw[i][nt], dR);
                    sb_r[i] += dR;
                }
            }
        }
        pp.mark(3);
        EEG_WAVE_SYNC();
#pragma unroll
        for (int i = 0; i < CT; ++i)
            if (own[i]) {
                lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, (wave + 4 * i) * 16, 2 * H, pf, lr, lg);
                lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, H + (wave + 4 * i) * 16, 2 * H, pf, lr, lg);
            }
        EEG_LDS_BARRIER();                                            // (2) P_m^T [dR|dU] complete
        pp.mark(4);

        // ---- GEMM2: dh = dhn + [P_m^T dG]_m (32 x M*2H) @ Wg^h^T (M*2H x H)
        mfma_nodes32<CT, KSG, REM4, 0, 0, false, false, REM4>(EG, KGP, lane, lr, lg, w2, dhn);
        pp.mark(5);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            dh[i][0] = dhn[i][0];
            dh[i][1] = dhn[i][1];
        }
    }

    // ---- epilogue: dh0 and the per-clip bias-gradient partial sums (fixed-order node reduction)
    __syncthreads();                                                // all waves done with EG
    float* red = EG;                                                // [3H][16] (+ REM4: [3H][4] partials of the remainder elements)
    float* red1 = EG + 3 * H * 16;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        if (own[i]) {
            const int col = (wave + 4 * i) * 16 + 4 * lg;
            if (dh0 != nullptr) {
                if (valid[0]) st4(dh0 + boff + node[0] * H + col, dh[i][0]);
                if (REM4) {
                    if (valid1) dh0[boff + node1 * H + (wave + 4 * i) * 16 + lr] = dh[i][1][0];
                } else if (valid[1]) {
                    st4(dh0 + boff + node[1] * H + col, dh[i][1]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[(0 * H + col + r) * 16 + lr] = sb_r[i][r];
                red[(1 * H + col + r) * 16 + lr] = sb_u[i][r];
                red[(2 * H + col + r) * 16 + lr] = sb_c[i][r];
            }
            if (REM4) {
                const int c1 = (wave + 4 * i) * 16 + lr;
                red1[(0 * H + c1) * 4 + lg] = sb1_r[i];
                red1[(1 * H + c1) * 4 + lg] = sb1_u[i];
                red1[(2 * H + c1) * 4 + lg] = sb1_c[i];
            }
        }
    }
    __syncthreads();
    for (int j = tid; j < 3 * H; j += 256) {
        float sacc = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) sacc += red[j * 16 + q];
        if (REM4) sacc += (red1[j * 4] + red1[j * 4 + 1]) + (red1[j * 4 + 2] + red1[j * 4 + 3]);
        dbias_part[(size_t)b * 3 * H + j] = sacc;
    }
    }   // clips of this workgroup
    pp.dump(probe, 8);
}

// ---- two waves per SIMD in the BPTT kernel (rnn_units = 64, at most 20 nodes, M <= 3) ---------------------------
// The BPTT step is a latency chain as well (dh -> dC,dU -> P^T dC -> GEMM1 -> dR -> P^T dR -> GEMM2 -> dh), but a good
// part of a step does not sit on it:
//   * HALF of its largest GEMM: GEMM2 contracts [P_m^T dR | P_m^T dU] with Wg^h^T, and dU is known right after the
//     blend backward (E1) -- it does not depend on GEMM1;
//   * everything of the elementwise steps that does not depend on the incoming gradient g: with
//     kC = (1-u) act'(c), kU = (h-c) u (1-u), hr1 = h r (1-r) the step is dC = g kC, dU = g kU, dh_part = g u,
//     dR = d(rh) hr1, dh_part += d(rh) r;
//   * the operand traffic (h_{t-1}, r, u, c, external gradient of the next step) and the per-clip bias sums.
// So 8 waves: wave w (role A, the chain, raised priority) owns column tile w of every H-wide quantity like
// seq_bwd_kernel -- the g-dependent multiplies, the three node mixes, GEMM1 -- and keeps only w1 (48 registers) and no
// operand loads at all; wave 4+w (role B, same SIMD) owns GEMM2 of column tile w (w2: 96 registers; the dU half runs
// while A does GEMM1 / dR / P^T dR, only the dR half is on the chain), fetches the operands of the NEXT step a step
// ahead, turns them into the five coefficient vectors and leaves them in a lane-linear LDS block that A reads with
// conflict-free 16-byte reads, adds the external gradient to its GEMM2 result before handing it to A (tile DP), and
// gathers the bias sums from the slot-0 tiles.  No register spills in either role: a scratch reload queues behind
// the outstanding prefetches in the memory pipe (keeping half of w2 in A spilled 105 registers and ran 2x slower;
// 15-30 spills cost 10-25 %).  The matrix pipe is shared, so no MFMA capacity is gained: the chain gets shorter and
// B's work fills A's waits.  M >= 4: w2 alone exceeds the 256 registers of a two-wave SIMD.
//   window 0 (after barrier 3 of step t+1)  A: g = dh_part + DP, dC, dU, P^T dC, P^T dU; takes hr1, r    B: sums dR(t+1)
//   window 1 (after barrier 1)               A: GEMM1, dR, P^T dR       B: coefficients(t-1) -> LDS, requests operands(t-2); GEMM2 dU half; sums dC, dU
//   window 2 (after barrier 2)               A: -                       B: GEMM2 dR half; DP = result + external gradient(t-1)
// SPEC (spectral form of the hoisted x-part, spec_common.h): the step's [dR|dU|dC] leaves as dYh = U^T dXW in the NODE-major layout
// (N, Sp, 3H) the grouped weight-gradient / input-gradient GEMMs read (row of (t, b) = t*B + b), instead
// of dXW -- role A mixes its column tile of dC, dU, dR with U^T in window 2, where it has nothing else to do (15 + 15 MFMAs beside
// role B's dR half), from the slot-0 columns it wrote itself.  A separate HBM pass over dXW (read 3H, write 3H per node row) is gone.
template <int H, int M, int NKS, bool PROBE = false, bool SPEC = false>
__global__ __launch_bounds__(512, 1) void seq_bwd2_kernel(
    const float* __restrict__ Hseq, const float* __restrict__ h0, const float* __restrict__ Rs,
    const float* __restrict__ Us, const float* __restrict__ Cs, const float* __restrict__ dHseq,
    const float* __restrict__ d_at_end, const float* __restrict__ d_at_len, const long long* __restrict__ lengths,
    const float* __restrict__ P, int p_batched, const float* __restrict__ b1p, const float* __restrict__ b2p,
    float* __restrict__ dXW, float* __restrict__ dh0, float* __restrict__ dbias_part, int T, int B, int N, int act,
    long long* probe, const float* __restrict__ spec_U = nullptr, float* __restrict__ dYh = nullptr, int spec_Sp = 0) {
    using G = SeqGeom<H, M>;
    static_assert(G::CT == 1 && NKS == 5, "one column tile per wave, second node tile on the 4x4x1 MFMA");
    constexpr int KAP = G::KAP, KS = G::KS, KGP = G::KGP, KSG = G::KSG, NCT = G::NCT, ROWS = 32, DPS = 20;
    // coefficient block of one (column tile, buffer): [kC, kU, u, hr1, r][64 lanes x 4] of nodes 0..15, one float4 {kC, kU, u, hr1}
    // per lane + one dword r per lane of the remainder element.  TWO buffers (step parity): role B fills the block of step t-1
    // in window 0 of step t, while role A reads the block of step t
    constexpr int kCoefTile = 5 * 256 + 256 + 64;
    PhaseProbe<PROBE> pp;
    pp.start();
    EEG_DYN_SMEM(sm);
    float* Pl = sm;
    float* EC = Pl + (M - 1) * kPFloats;    // [32][KAP]  slot 0 = dC, slots m = P_m^T dC
    float* EG = EC + ROWS * KAP;            // [32][KGP]  slot m = [P_m^T dR | P_m^T dU]
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6, role = wave8 >> 2, ct = wave8 & 3;
    const int lr = lane & 15, lg = lane >> 4;
    float* DP = EG + ROWS * KGP + ct * (20 * DPS);                     // [20][DPS] GEMM2 result + external gradient, tile ct
    float* CF0 = EG + ROWS * KGP + 4 * 20 * DPS + ct * kCoefTile;      // this column tile's coefficient block, buffer 0 (buffer 1: + 4 * kCoefTile)
    constexpr int kZero = ROWS * (KAP + KGP) + 4 * 20 * DPS + 2 * 4 * kCoefTile;   // floats cleared per clip
    const int node[2] = {lr, 16 + lr};
    const bool valid[2] = {lr < N, 16 + lr < N};
    const int nodec[2] = {valid[0] ? lr : N - 1, valid[1] ? 16 + lr : N - 1};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int col = ct * 16 + 4 * lg;
    // remainder nodes 16..19, one value per lane (mfma_nodes32 L1): lane (lr, lg) <-> node 16 + lg, column ct*16 + lr
    const int node1 = 16 + lg, col1 = ct * 16 + lr;
    const bool valid1 = node1 < N;
    const int odp0 = lr * DPS + 4 * lg, odp1 = node1 * DPS + lr;               // lane <-> lane hand-over ([20][DPS] = [node][col])
    // coefficient slots of the remainder element: one float4 {kC, kU, u, hr1} per lane in the nt = 1 slot of coefficient 0, r in
    // the nt = 1 slot of coefficient 1 (lane-linear dwords); the nt = 1 slots of coefficients 2..4 are unused
    // this lane's slots inside a block: float4 k of nodes 0..15 at 256*k + 4*lane, the remainder float4 at 1280 + 4*lane, its r at 1536 + lane
    auto cf_buf = [&](int t) { return CF0 + (t & 1) * (4 * kCoefTile); };

    if (role == 1) {
        // ================= role B =================
        float w2[1][KSG];
#pragma unroll
        for (int ks = 0; ks < KSG; ++ks) w2[0][ks] = b2p[((size_t)ks * NCT + ct) * 64 + lane];
        const int oh0 = nodec[0] * H + col, oh1 = (valid1 ? node1 : N - 1) * H + col1;
        const int orow[2] = {lr, valid[1] ? 16 + lr : 16};             // bias sums, nt = 1: only lanes with a real node add
        for (int b = blockIdx.x; b < B; b += gridDim.x) {
            __syncthreads();
            for (int e = tid; e < kZero; e += 512) EC[e] = 0.f;
            lds_load_polys(Pl, P, p_batched ? b : 0, M, N);
            int t_len = -1;                                             // same clamp as gather_last_kernel
            if (d_at_len != nullptr) {
                t_len = lengths != nullptr ? (int)lengths[b] - 1 : T - 1;
                t_len = t_len < 0 ? 0 : (t_len >= T ? T - 1 : t_len);
            }
            const size_t tstride = (size_t)B * N * H, boff = (size_t)b * N * H;
            f32x4 nh, nr, nu, nc, ng;                                   // nodes 0..15: four columns per lane
            float nh1, nr1, nu1, nc1, ng1;                              // node 16 + lg: one column per lane
            // operands through buffer descriptors (one VGPR offset per node tile, the step offset in an SGPR): no 64-bit
            // per-lane address arithmetic on the VALU, which shares its ALUs with the fp32 MFMAs of the other role
            const wbuf_t bH = make_wbuf(Hseq), bH0 = make_wbuf(h0 != nullptr ? h0 : Hseq), bR = make_wbuf(Rs), bU = make_wbuf(Us),
                         bC = make_wbuf(Cs), bG = make_wbuf(dHseq != nullptr ? dHseq : Hseq);
            auto fetch = [&](int t) {
                const unsigned so = (unsigned)((size_t)t * tstride + boff);
                nh = t > 0 ? wbuf_ld4(bH, oh0, so - (unsigned)tstride) : (h0 != nullptr ? wbuf_ld4(bH0, oh0, (unsigned)boff) : zero4);
                nr = wbuf_ld4(bR, oh0, so);
                nu = wbuf_ld4(bU, oh0, so);
                nc = wbuf_ld4(bC, oh0, so);
                nh1 = t > 0 ? wbuf_ld(bH, oh1, so - (unsigned)tstride) : (h0 != nullptr ? wbuf_ld(bH0, oh1, (unsigned)boff) : 0.f);
                nr1 = wbuf_ld(bR, oh1, so);
                nu1 = wbuf_ld(bU, oh1, so);
                nc1 = wbuf_ld(bC, oh1, so);
                f32x4 g = dHseq != nullptr ? wbuf_ld4(bG, oh0, so) : zero4;
                float g1 = dHseq != nullptr ? wbuf_ld(bG, oh1, so) : 0.f;
                if (d_at_end != nullptr && t == T - 1) { g += ld4(d_at_end + boff + oh0); g1 += d_at_end[boff + oh1]; }
                if (t == t_len) { g += ld4(d_at_len + boff + oh0); g1 += d_at_len[boff + oh1]; }
                ng = g;
                ng1 = g1;
            };
            // the five coefficient vectors of one step from its operands -> this lane's LDS slots; the external gradient
            // of that step is kept (gx) and added to the GEMM2 result that becomes its incoming gradient
            f32x4 gx;
            float gx1;
            auto coef = [&](int t_of) {                                 // operands in registers belong to step t_of
                float* CF = cf_buf(t_of) + 4 * lane;
                {
                    const f32x4 h = nh, u = nu, c = nc, r = nr;
                    f32x4 kC, kU, hr1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float one_u = 1.f - u[e];
                        kC[e] = act == 0 ? one_u * (1.f - c[e] * c[e]) : (c[e] > 0.f ? one_u : 0.f);
                        kU[e] = (h[e] - c[e]) * u[e] * one_u;
                        hr1[e] = h[e] * r[e] * (1.f - r[e]);
                    }
                    st4(CF + 0 * 256, kC);
                    st4(CF + 1 * 256, kU);
                    st4(CF + 2 * 256, u);
                    st4(CF + 3 * 256, hr1);
                    st4(CF + 4 * 256, r);
                    gx = ng;
                }
                {
                    const float one_u = 1.f - nu1;
                    const float kC1 = act == 0 ? one_u * (1.f - nc1 * nc1) : (nc1 > 0.f ? one_u : 0.f);
                    st4(CF + 5 * 256, (f32x4){kC1, (nh1 - nc1) * nu1 * one_u, nu1, nh1 * nr1 * (1.f - nr1)});
                    CF[6 * 256 - 3 * lane] = nr1;                       // (block + 1536 + lane)
                    gx1 = ng1;
                }
            };
            fetch(T - 1);
            __syncthreads();                                            // tiles cleared
            coef(T - 1);                                                // first step: its coefficients, DP = external gradient
            if (T > 1) fetch(T - 2);
            st4(DP + odp0, gx);
            DP[odp1] = gx1;
            f32x4 sb_r = zero4, sb_u = zero4, sb_c = zero4;
            EEG_LDS_BARRIER();                                          // (3) of an imaginary step T
            for (int t = T - 1; t >= 0; --t) {
                if (t < T - 1) {                                        // window 0: dR of the step before is still in place
                    sb_r += ld4(EG + lds_sw(orow[0], col, KGP));
                    if (valid[1]) sb_r += ld4(EG + lds_sw(orow[1], col, KGP));
                }
                // window 0 (role A: elementwise work and node mixes, no GEMM on the pipe): the coefficients of step t-1 go to the
                // OTHER buffer -- VALU instructions issued beside a wave that streams MFMAs wait out a whole MFMA each
                if (t > 0) {
                    coef(t - 1);                                        // (operands requested a step ago)
                    if (t > 1) fetch(t - 2);                            // ... and the same registers request step t-2
                }
                EEG_LDS_BARRIER();                                      // (1) P_m^T dC and P_m^T dU complete
                f32x4 acc[1][2] = {{zero4, zero4}};
                mfma_nodes32<1, KSG, true, 0, 2, true, true, true>(EG, KGP, lane, lr, lg, w2, acc);     // dU half: off the chain
                sb_c += ld4(EC + lds_sw(orow[0], col, KAP));
                sb_u += ld4(EG + lds_sw(orow[0], H + col, KGP));
                if (valid[1]) {
                    sb_c += ld4(EC + lds_sw(orow[1], col, KAP));
                    sb_u += ld4(EG + lds_sw(orow[1], H + col, KGP));
                }
                EEG_LDS_BARRIER();                                      // (2) P_m^T dR complete
                EEG_SETPRIO(3);                                         // the dR half is on the chain
                mfma_nodes32<1, KSG, true, 0, 1, true, true, true>(EG, KGP, lane, lr, lg, w2, acc);
                EEG_SETPRIO(0);
                if (t > 0) { acc[0][0] += gx; acc[0][1][0] += gx1; }
                st4(DP + odp0, acc[0][0]);
                DP[odp1] = acc[0][1][0];
                EEG_LDS_BARRIER();                                      // (3) DP complete
            }
            sb_r += ld4(EG + lds_sw(orow[0], col, KGP));                // dR of the last step (t = 0)
            if (valid[1]) sb_r += ld4(EG + lds_sw(orow[1], col, KGP));
            __syncthreads();                                            // all waves done with EG
            float* red = EG;                                            // [3H][16]: fixed-order node reduction
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[(0 * H + col + r) * 16 + lr] = sb_r[r];
                red[(1 * H + col + r) * 16 + lr] = sb_u[r];
                red[(2 * H + col + r) * 16 + lr] = sb_c[r];
            }
            __syncthreads();
            for (int j = tid - 256; j < 3 * H; j += 256) {
                float sacc = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) sacc += red[j * 16 + q];
                dbias_part[(size_t)b * 3 * H + j] = sacc;
            }
        }
        return;
    }
    // ================= role A =================
    float w1[1][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) w1[0][ks] = b1p[((size_t)ks * NCT + ct) * 64 + lane];
    for (int b = blockIdx.x; b < B; b += gridDim.x) {               // clips of this workgroup (see seq_fwd_kernel)
    __syncthreads();                                                // previous clip: the bias reduction has read EG
    for (int e = tid; e < kZero; e += 512) EC[e] = 0.f;
    lds_load_polys(Pl, P, p_batched ? b : 0, M, N);
    __syncthreads();                                                // tiles cleared
    EEG_SETPRIO(3);                                                 // windows 0 and 1: the chain issues first
    float pf[poly_slots<M, NKS>()][NKS];
    load_poly_frags<M, NKS, true>(Pl, pf, lr, lg);
    float uf[NKS], u4[NKS];                                         // SPEC: this lane's fragments of U
    if constexpr (SPEC) load_spec_frags<NKS>(spec_U, N, lane, uf, u4);
    const wbuf_t bY = make_wbuf(SPEC ? dYh : dXW);
    const unsigned oy0 = (unsigned)lr * (unsigned)spec_Sp * (3 * H) + col, oy1 = (unsigned)node1 * (unsigned)spec_Sp * (3 * H) + col1;
    const int oxw0 = node[0] * (3 * H) + col, oxw1 = node1 * (3 * H) + col1;
    const int lc1 = lds_sw(node1, col1, KAP), lg1 = lds_sw(node1, col1, KGP), lu1 = lds_sw(node1, H + col1, KGP);
    const size_t boff = (size_t)b * N * H;
    f32x4 dhn = zero4;                                              // A's elementwise part of dh (nodes 0..15)
    float dhn1 = 0.f;                                               // ... of node 16 + lg
    EEG_LDS_BARRIER();                                              // (3) of an imaginary step T: first coefficients, DP
    const wbuf_t bX = make_wbuf(dXW);
    pp.stamp(1);
    pp.last = PROBE ? cycle_now() : 0;
    ClockSample cs;
    const bool cs_me = !PROBE && b == 0 && tid == 0;
    cs.begin(probe == nullptr ? nullptr : probe + 2, cs_me);
    for (int t = T - 1; t >= 0; --t) {
        const unsigned sx = (unsigned)(((size_t)t * B + b) * N * (3 * H));
        // ---- E1: g = A's elementwise part + role B's GEMM2 of the step before (+ external gradient, added by B)
        f32x4 hr1, rg;
        float hr1_1, rg1;
        {
            const float* CF = cf_buf(t) + 4 * lane;                 // block of step t (role B is filling the other one)
            const f32x4 g = valid[0] ? dhn + ld4(DP + odp0) : zero4;
            const f32x4 dC = g * ld4(CF + 0 * 256), du_ = g * ld4(CF + 1 * 256);
            st4(EC + lds_sw(lr, col, KAP), dC);                     // zeros on padding nodes
            st4(EG + lds_sw(lr, H + col, KGP), du_);
            if (!SPEC && valid[0]) {
                wbuf_st4(bX, oxw0 + 2 * H, sx, dC);
                wbuf_st4(bX, oxw0 + H, sx, du_);
            }
            dhn = g * ld4(CF + 2 * 256);
            hr1 = ld4(CF + 3 * 256);
            rg = ld4(CF + 4 * 256);
            // node 16 + lg, column ct*16 + lr: {kC, kU, u, hr1} in one 16-byte read
            const float g1 = valid1 ? dhn1 + DP[odp1] : 0.f;
            const f32x4 k1 = ld4(CF + 5 * 256);
            const float dC1 = g1 * k1[0], du1 = g1 * k1[1];
            EC[lc1] = dC1;
            EG[lu1] = du1;
            if (!SPEC && valid1) {
                wbuf_st1(bX, oxw1 + 2 * H, sx, dC1);
                wbuf_st1(bX, oxw1 + H, sx, du1);
            }
            dhn1 = g1 * k1[2];
            hr1_1 = k1[3];
            rg1 = CF[6 * 256 - 3 * lane];
        }
        pp.mark(0);
        EEG_WAVE_SYNC();
        lds_diffuse_tile<M, NKS, ROWS>(EC, KAP, ct * 16, H, pf, lr, lg);
        lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, H + ct * 16, 2 * H, pf, lr, lg);
        pp.mark(7);
        EEG_LDS_BARRIER();                                          // (1) P_m^T dC and P_m^T dU complete
        pp.mark(1);
        // ---- GEMM1: d(r*h) = [P_m^T dC]_m (32 x M*H) @ Wc^h^T
        f32x4 acc[1][2] = {{zero4, zero4}};
        mfma_nodes32<1, KS, true, 0, 0, false, false, true>(EC, KAP, lane, lr, lg, w1, acc);   // (TWOPASS here: seq_bwd +2.5 %, measured)
        pp.mark(2);
        {
            const f32x4 drh = acc[0][0];                            // exact 0 on padding nodes
            const f32x4 dR = drh * hr1;
            dhn += drh * rg;
            st4(EG + lds_sw(lr, col, KGP), dR);
            if (!SPEC && valid[0]) wbuf_st4(bX, oxw0, sx, dR);
            const float drh1 = acc[0][1][0], dR1 = drh1 * hr1_1;
            dhn1 += drh1 * rg1;
            EG[lg1] = dR1;
            if (!SPEC && valid1) wbuf_st1(bX, oxw1, sx, dR1);
        }
        pp.mark(3);
        EEG_WAVE_SYNC();
        lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, ct * 16, 2 * H, pf, lr, lg);
        EEG_LDS_BARRIER();                                          // (2) P_m^T dR complete
        pp.mark(4);
        EEG_SETPRIO(0);                                             // window 2 belongs to role B's half GEMM
        if constexpr (SPEC) {
            // dYh = U^T [dR | dU | dC] of this column tile, from the slot-0 columns this wave wrote itself (complete for all nodes)
            const float* const tl[3] = {EG, EG, EC};
            const int strd[3] = {KGP, KGP, KAP}, scol[3] = {ct * 16, H + ct * 16, ct * 16}, gcol[3] = {0, H, 2 * H};
            const unsigned row = (unsigned)t * (unsigned)B + (unsigned)b;
            EEG_WAVE_SYNC();
            spec_mix_tiles_out<NKS, 3>(tl, strd, scol, gcol, uf, u4, lr, lg, bY, oy0, oy1, row * (3 * H), valid[0], valid1);
        }
        EEG_LDS_BARRIER();                                          // (3) role B's GEMM2 is in DP
        EEG_SETPRIO(3);
        pp.mark(6);
    }
    cs.end(probe == nullptr ? nullptr : probe + 2, cs_me);
    pp.stamp(2);
    // ---- epilogue: dh0; role B reduces the bias sums
    __syncthreads();                                                // all waves done with EG
    if (dh0 != nullptr) {
        if (valid[0]) st4(dh0 + boff + lr * H + col, dhn + ld4(DP + odp0));
        if (valid1) dh0[boff + node1 * H + col1] = dhn1 + DP[odp1];
    }
    __syncthreads();
    }   // clips of this workgroup
    pp.dump(probe, 8);
}

}  // namespace eeg
