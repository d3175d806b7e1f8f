// Persistent decoder forward (model/model.py:160-204, DCGRUDecoder.forward): ONE launch runs all T_out autoregressive
// steps -- every layer's DCGRU cell, the projection and the feedback of the prediction -- for the clips it owns.
//
// Why: the input of step t+1 is the projection of step t, so nothing of the decoder's forward can be hoisted over
// time; launched per step it is ~7 dependent launches per step (input diffusion, x-part GEMM, recurrent step per
// layer, projection, copy), each over only B*N rows, and every T = 1 recurrent launch re-loads 147-246 KB of
// register-resident weights per workgroup (round 1: dec_seq_fwd 0.20, dec_gemm_nn 0.31 of their roofs).  Clips are
// independent, so a workgroup can keep ITS clip's whole decoder state on chip instead:
//   * the hidden states h^l and their hop rows P_m h^l stay in LDS tiles for all steps (as in kernels_seq.h), the
//     step input (previous prediction) and its hop rows too;
//   * NO weight lives in registers: every GEMM streams its weight fragments from L2 (the packs of kernels_pack.h,
//     read with coalesced 256-byte wave loads, a group of k-steps ahead of the MFMAs that consume them).  All
//     workgroups stream the same few hundred KB, which the 4 MB L2 of every XCD holds;
//   * the x-part of a cell (input hops x W^x + bias) is not a separate GEMM: it is accumulated into the SAME MFMA
//     accumulators as the h-part, gate and candidate columns together (they share the input fragments);
//   * everything the backward needs (r, u, c, r*h, h, hop planes of h and r*h, hop planes of the step input, the
//     step inputs themselves) is written in the layout of eeg_dcrnn_decoder_fwd's `saved` block, so the existing
//     backward operator is unchanged.
// Wave w of 4 owns column tile w of r, u, c and h (64 units).  At most 20 nodes (second node tile on v_mfma_f32_4x4x1).
#pragma once
#include "common.h"
#include "kernels_seq.h"
#include "lds_diffuse.h"

namespace eeg {

struct DecLayerPtrs {
    const float *bx, *bias, *bhg, *bhc;                       // weight packs of the layer's cell (kernels_pack.h)
    float *hext, *rs, *us, *cs, *rhs, *hpl, *rpl;             // saved for the backward (decoder `saved` layout)
};
struct DecFwdArgs {
    DecLayerPtrs l[4];
    const float* P;
    const float* targets;         // (T,B,N,Dout), read where teacher_mask says so (may be NULL when the mask is 0)
    const float *ppack, *pbias;   // projection: fragment pack (K = H, nct_o column tiles) and zero-padded bias
    float *out, *xin, *planes0;   // (T,B,N,Dout) predictions, step inputs, and the hop planes of the step inputs (M-1 planes)
    size_t planes0_stride;        // floats between two planes of planes0
    size_t hplane_stride;         // floats between two planes of hpl / rpl = (T+1)*B*N*H
    unsigned long long teacher_mask;   // bit t: step t+1 is fed targets[t] instead of out[t] (model.py:194-200)
    int p_batched, T, B, N, Dout, L, act;
};

// ---- GEMMs with streamed weights -------------------------------------------------------------------------------
// acc[i][nt] += (weight tile wt[i])^T x X^T, issued transposed like mfma_nodes32 (lane: node lr / 16 + lr, 4 consecutive
// columns), remainder nodes 16..19 on v_mfma_f32_4x4x1 with the hand-over through `scratch` (NT * kRemTile floats).
// Plain K order (pack index ks = k/4, k = slot*KPP*4 + f): the operand tile has `nslots` hop slots of `slotw` columns of
// which the first 4*KPP are real; SWZ: the tile is an XOR-swizzled state tile (lds_sw), else a plain [rows][stride] one.
// The weights of group g+1 (D k-steps x NT tiles, one coalesced dword per lane each) are requested before the MFMAs
// of group g.  KPP % D == 0.
template <int NT, int D, bool SWZ>
__device__ __forceinline__ void gemm_stream_plain(const float* __restrict__ tile, int stride, int slotw, int kpp, int nslots,
                                                  const float* __restrict__ wp, int nct_total, const int (&wt)[NT],
                                                  int lane, int lr, int lg, f32x4 (&acc)[NT][2], float* scratch) {
    const int ngroups = nslots * kpp / D;
    const int row1 = 16 + (lane & 3);
    f32x4 rem[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rem[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wl = wp + lane;
    auto wload = [&](int g, float (&w)[D][NT]) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int i = 0; i < NT; ++i) w[d][i] = wl[((size_t)(g * D + d) * nct_total + wt[i]) * 64];
    };
    auto xload = [&](int slot, int f0, float (&x0)[D], float (&x1)[D]) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int col = slot * slotw + 4 * (f0 + d) + lg;
            x0[d] = tile[SWZ ? lds_sw(lr, col, stride) : lr * stride + col];
            x1[d] = tile[SWZ ? lds_sw(row1, col, stride) : row1 * stride + col];
        }
    };
    auto mac = [&](const float (&w)[D][NT], const float (&x0)[D], const float (&x1)[D]) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                acc[i][0] = mfma16(w[d][i], x0[d], acc[i][0]);
                rem[i][d & 3] = mfma4(x1[d], w[d][i], rem[i][d & 3]);
            }
    };
    float wa[D][NT], wb[D][NT], x0[D], x1[D];
    int slot = 0, f0 = 0;
    wload(0, wa);
    for (int g = 0; g < ngroups; g += 2) {
        if (g + 1 < ngroups) wload(g + 1, wb);
        xload(slot, f0, x0, x1);
        EEG_SCHED_FENCE();
        mac(wa, x0, x1);
        f0 += D;
        if (f0 == kpp) { f0 = 0; ++slot; }
        if (g + 1 < ngroups) {
            if (g + 2 < ngroups) wload(g + 2, wa);
            xload(slot, f0, x0, x1);
            EEG_SCHED_FENCE();
            mac(wb, x0, x1);
            f0 += D;
            if (f0 == kpp) { f0 = 0; ++slot; }
        }
    }
    // remainder hand-over (see mfma_nodes32)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            scratch[i * kRemTile + lg * 80 + r * 16 + lr] = (rem[i][0][r] + rem[i][1][r]) + (rem[i][2][r] + rem[i][3][r]);
    EEG_WAVE_SYNC();
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const float* q = scratch + i * kRemTile + (lr & 3) * 16 + 4 * lg;
        const f32x4 s = (ld4(q) + ld4(q + 80)) + (ld4(q + 160) + ld4(q + 240));
        if (lr < 4) acc[i][1] += s;
    }
    EEG_WAVE_SYNC();
}

// Same for the recurrent packs (quad-permuted K order, ds_read_b128 fragments of a swizzled state tile): NQ quads
// (compile time), weights requested PD quads ahead.
template <int NT, int NQ, int PD>
__device__ __forceinline__ void gemm_stream_quad(const float* __restrict__ tile, int stride, const float* __restrict__ wp,
                                                 int nct_total, const int (&wt)[NT], int lane, int lr, int lg,
                                                 f32x4 (&acc)[NT][2], float* scratch) {
    const int s0 = lg ^ sigma4(lr), s1 = lg ^ sigma4(lane & 3);
    const float* p0 = tile + lr * stride;
    const float* p1 = tile + (16 + (lane & 3)) * stride;
    auto frag = [&](const float* rowp, int sx, int q) {
        return *reinterpret_cast<const float4*>(rowp + 64 * (q >> 2) + 4 * ((4 * (q & 3)) ^ sx));
    };
    const float* wl = wp + lane;
    float w[PD + 1][4][NT];
    auto wload = [&](int q, float (&dst)[4][NT]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) dst[j][i] = wl[((size_t)(4 * q + j) * nct_total + wt[i]) * 64];
    };
    f32x4 rem[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rem[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < PD && q < NQ; ++q) wload(q, w[q % (PD + 1)]);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q + PD < NQ) wload(q + PD, w[(q + PD) % (PD + 1)]);
        const float4 a0 = frag(p0, s0, q), a1 = frag(p1, s1, q);
        const float x0[4] = {a0.x, a0.y, a0.z, a0.w}, x1[4] = {a1.x, a1.y, a1.z, a1.w};
        EEG_SCHED_FENCE();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                acc[i][0] = mfma16(w[q % (PD + 1)][j][i], x0[j], acc[i][0]);
                rem[i][j] = mfma4(x1[j], w[q % (PD + 1)][j][i], rem[i][j]);
            }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            scratch[i * kRemTile + lg * 80 + r * 16 + lr] = (rem[i][0][r] + rem[i][1][r]) + (rem[i][2][r] + rem[i][3][r]);
    EEG_WAVE_SYNC();
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const float* q = scratch + i * kRemTile + (lr & 3) * 16 + 4 * lg;
        const f32x4 s = (ld4(q) + ld4(q + 80)) + (ld4(q + 160) + ld4(q + 240));
        if (lr < 4) acc[i][1] += s;
    }
    EEG_WAVE_SYNC();
}

constexpr int kDecRows = 20;     // node rows of the LDS tiles (montages of at most 20 nodes)

// LDS floats of dec_fwd_persist_kernel<64, M>
__host__ __device__ constexpr size_t dec_fwd_lds_floats(int M, int L, int Dout) {
    const int H = 64, KAP = M * H, XS = M * round_up(Dout, 16);
    return (size_t)(M - 1) * kPFloats + (size_t)L * kDecRows * KAP + (size_t)kDecRows * (XS > KAP ? XS : KAP) + 4 * 3 * kRemTile;
}

// DX = k-steps per weight group of the layer-0 x-part ((Dout/4) % DX == 0; the larger, the further ahead the weights are requested).
template <int H, int M, int DX>
__global__ __launch_bounds__(256, 1) void dec_fwd_persist_kernel(DecFwdArgs a) {
    static_assert(H == 64, "one column tile per wave");
    constexpr int NKS = 5, ROWS = kDecRows, KAP = M * H, NCT = H / 16, NGT = 2 * NCT, NQ = M * H / 16;
    EEG_DYN_SMEM(sm);
    const int T = a.T, B = a.B, N = a.N, Dout = a.Dout, L = a.L, act = a.act;
    const int FP = round_up(Dout, 16), XS = M * FP, XK = XS > KAP ? XS : KAP;
    float* Pl = sm;
    float* A0 = Pl + (M - 1) * kPFloats;            // L state tiles [ROWS][KAP]: slot 0 = h^l, slots m = P_m h^l
    float* XA = A0 + L * ROWS * KAP;                // step-input tile X0 [ROWS][XS] (plain)  |  r*h tile A2 [ROWS][KAP] (swizzled)
    float* RS0 = XA + ROWS * XK;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), lr = lane & 15, lg = lane >> 4;
    float* RS = RS0 + wave * (3 * kRemTile);
    const int ct = wave, col = ct * 16 + 4 * lg;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int nct_o = ceil_div(Dout, 16);
    const size_t xstep = (size_t)B * N * Dout;

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < L * ROWS * KAP + ROWS * XK; e += 256) A0[e] = 0.f;
        lds_load_polys(Pl, a.P, a.p_batched ? b : 0, M, N);
        __syncthreads();
        float pf[poly_slots<M, NKS>()][NKS];
        load_poly_frags<M, NKS, false>(Pl, pf, lr, lg);
        const int node[2] = {lr, 16 + lr};
        const bool valid[2] = {lr < N, 16 + lr < N};
        const int nodec[2] = {valid[0] ? lr : N - 1, valid[1] ? 16 + lr : N - 1};
        const int oh[2] = {nodec[0] * H + col, nodec[1] * H + col};
        // initial states (encoder finals; the host has copied them into hext slot 0) and their hop rows (hpl slot 0)
        for (int l = 0; l < L; ++l) {
            float* Al = A0 + l * ROWS * KAP;
            for (int e = tid; e < N * H; e += 256) Al[lds_sw(e / H, e % H, KAP)] = a.l[l].hext[(size_t)b * N * H + e];
        }
        __syncthreads();
        for (int l = 0; l < L; ++l)
            lds_diffuse_tile<M, NKS, ROWS>(A0 + l * ROWS * KAP, KAP, ct * 16, H, pf, lr, lg,
                                           a.l[l].hpl + (size_t)b * N * H, a.hplane_stride, N);
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const size_t s = (size_t)t * B + b;
            // ---- hop rows of the step input (X0 slot 0 holds it: zeros at t = 0, GO symbol) -> slots 1..M-1, and to planes0
            lds_diffuse_tiles<false>(XA, XS, 0, FP, FP, FP, Pl, M, N, ROWS);
            __syncthreads();
            for (int m1 = 0; m1 < M - 1; ++m1) {
                float* dst = a.planes0 + (size_t)m1 * a.planes0_stride + s * N * Dout;
                for (int e = tid; e < N * (Dout / 4); e += 256) {
                    const int n = e / (Dout / 4), c4 = e % (Dout / 4);
                    *reinterpret_cast<float4*>(dst + n * Dout + 4 * c4) =
                        *reinterpret_cast<const float4*>(XA + n * XS + (m1 + 1) * FP + 4 * c4);
                }
            }
            for (int l = 0; l < L; ++l) {
                const DecLayerPtrs& lp = a.l[l];
                float* Al = A0 + l * ROWS * KAP;
                float* A2 = XA;
                f32x4 ag[3][2];          // pre-activations of this wave's r, u, c tiles: x-part + bias, then + h-part
                const int wt3[3] = {ct, NCT + ct, 2 * NCT + ct};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const f32x4 bv = ld4(lp.bias + wt3[i] * 16 + 4 * lg);
                    ag[i][0] = bv;
                    ag[i][1] = lr < 4 ? bv : zero4;
                }
                if (l == 0)
                    gemm_stream_plain<3, DX, false>(XA, XS, FP, Dout / 4, M, lp.bx, 3 * NCT, wt3, lane, lr, lg, ag, RS);
                else
                    gemm_stream_plain<3, 16, true>(A0 + (l - 1) * ROWS * KAP, KAP, H, H / 4, M, lp.bx, 3 * NCT, wt3, lane, lr, lg, ag, RS);
                __syncthreads();                                     // layer 0: every wave has read X0 (A2 aliases it)
                // gate h-part: hops(h^l) x Wg^h
                {
                    f32x4 g2[2][2] = {{ag[0][0], ag[0][1]}, {ag[1][0], ag[1][1]}};
                    const int wt2[2] = {ct, NCT + ct};
                    gemm_stream_quad<2, NQ, (NQ < 6 ? NQ : 6)>(Al, KAP, lp.bhg, NGT, wt2, lane, lr, lg, g2, RS);
                    ag[0][0] = g2[0][0]; ag[0][1] = g2[0][1]; ag[1][0] = g2[1][0]; ag[1][1] = g2[1][1];
                }
                f32x4 ug[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 rg, u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        rg[r] = sigmoidf_(ag[0][nt][r]);
                        u[r] = sigmoidf_(ag[1][nt][r]);
                    }
                    ug[nt] = u;
                    f32x4 rh = rg * ld4(Al + lds_sw(nt == 0 ? lr : 16 + (lr & 3), col, KAP));
                    rh = valid[nt] ? rh : zero4;
                    if (nt == 0 || lr < 4) st4(A2 + lds_sw(nt == 0 ? lr : 16 + lr, col, KAP), rh);
                    if (valid[nt]) {
                        st4(lp.rs + s * N * H + oh[nt], rg);
                        st4(lp.rhs + s * N * H + oh[nt], rh);
                        st4(lp.us + s * N * H + oh[nt], u);
                    }
                }
                EEG_WAVE_SYNC();
                lds_diffuse_tile<M, NKS, ROWS>(A2, KAP, ct * 16, H, pf, lr, lg, lp.rpl + s * N * H, a.hplane_stride, N);
                __syncthreads();                                     // hops(r*h) complete
                // candidate h-part: hops(r*h) x Wc^h
                {
                    f32x4 c1[1][2] = {{ag[2][0], ag[2][1]}};
                    const int wt1[1] = {ct};
                    gemm_stream_quad<1, NQ, (NQ < 10 ? NQ : 10)>(A2, KAP, lp.bhc, NCT, wt1, lane, lr, lg, c1, RS);
                    ag[2][0] = c1[0][0]; ag[2][1] = c1[0][1];
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int row = nt == 0 ? lr : 16 + (lr & 3);
                    const f32x4 u = ug[nt], h = ld4(Al + lds_sw(row, col, KAP));
                    f32x4 c, hn;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pre = ag[2][nt][r];
                        c[r] = act == 0 ? tanhf_(pre) : fmaxf(pre, 0.f);
                        hn[r] = u[r] * h[r] + (1.f - u[r]) * c[r];
                    }
                    hn = valid[nt] ? hn : zero4;
                    if (nt == 0 || lr < 4) st4(Al + lds_sw(nt == 0 ? lr : 16 + lr, col, KAP), hn);
                    if (valid[nt]) {
                        st4(lp.hext + (s + B) * N * H + oh[nt], hn);             // hext slot t+1
                        st4(lp.cs + s * N * H + oh[nt], c);
                    }
                }
                EEG_WAVE_SYNC();
                lds_diffuse_tile<M, NKS, ROWS>(Al, KAP, ct * 16, H, pf, lr, lg, lp.hpl + (s + B) * N * H, a.hplane_stride, N);
                __syncthreads();                                     // h^l of this step and its hop rows complete
            }
            // ---- projection (model.py:188-190) and the next step's input (model.py:194-200)
            {
                float* Atop = A0 + (L - 1) * ROWS * KAP;
                const bool tf = ((a.teacher_mask >> t) & 1ull) != 0;
                for (int j0 = wave; j0 < nct_o; j0 += 8) {              // this wave's tiles j0 and j0 + 4
                    const bool two = j0 + 4 < nct_o;
                    f32x4 po[2][2];
                    const int wt2[2] = {j0, two ? j0 + 4 : j0};
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f32x4 bv = ld4(a.pbias + wt2[i] * 16 + 4 * lg);
                        po[i][0] = bv;
                        po[i][1] = lr < 4 ? bv : zero4;
                    }
                    gemm_stream_plain<2, 16, true>(Atop, KAP, H, H / 4, 1, a.ppack, nct_o, wt2, lane, lr, lg, po, RS);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (i == 1 && !two) continue;
                        const int c0 = wt2[i] * 16 + 4 * lg;
                        if (c0 >= Dout) continue;                       // Dout % 4 == 0: whole float4 pieces
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            if (!valid[nt]) continue;
                            const size_t o = (s * N + node[nt]) * Dout + c0;
                            st4(a.out + o, po[i][nt]);
                            if (t + 1 < T) {
                                const f32x4 nxt = tf ? ld4(a.targets + o) : po[i][nt];
                                st4(a.xin + o + xstep, nxt);
                                st4(XA + node[nt] * XS + c0, nxt);      // X0 slot 0 of the next step
                            }
                        }
                    }
                }
            }
            __syncthreads();                                         // X0 slot 0 of the next step complete
        }
    }
}

}  // namespace eeg

namespace eeg {

// ---- persistent decoder backward ---------------------------------------------------------------------------------
// The BPTT mirror of dec_fwd_persist_kernel: ONE launch walks the T_out steps backwards for the clips it owns -- per
// step the projection transpose, every layer's cell backward (the step of seq_bwd_kernel: blend backward, P^T dC,
// GEMM1, dR, P^T [dR|dU], GEMM2), the input gradient of the layer (dXW x W^x^T, then the adjoint node mix) which feeds
// the layer below or, through the autoregressive feedback, the previous step's output gradient -- with every weight
// streamed from L2 (packs b1, b2, bxt of kernels_pack.h and the transposed projection pack).  On chip per clip: the
// dC / [dR|dU] tiles with their adjoint hop rows (the dead hop slots of the [dR|dU] tile hold Z = dXW W^x^T between
// GEMM2 and the adjoint mix), the output-gradient tile, the recurrent gradients dh^l (lane-linear LDS slots).  It emits
// dXW of every (layer, step) -- the hoisted parameter-gradient GEMMs, the bias column sums and the projection gradients
// stay as they are -- the total output gradients dOtot and dh0.  At most 20 nodes, 64 units.
struct DecBwdLayerPtrs {
    const float *b1, *b2, *bxt;                      // weight packs
    const float *hext, *rs, *us, *cs;                // saved by the forward
    float* dxw;                                      // (T,B,N,3H) out
};
struct DecBwdArgs {
    DecBwdLayerPtrs l[4];
    const float* P;
    const float* tpack;           // projection, transposed role: K = Dout, H/16 column tiles
    const float* dOut;            // (T,B,N,Dout) loss gradient
    float* dOtot;                 // (T,B,N,Dout) total gradient of out_t (loss + feedback)
    float* dh0;                   // (L,B,N,H)
    unsigned long long feeds_mask;     // bit t: out_t is the input of step t+1 (no teacher forcing there, t+1 < T)
    int p_batched, T, B, N, Dout, L, act;
};

// Z (M*Fin columns, Fin = Dout for layer 0, H above) lives in the hop slots of the [dR|dU] tile when they are wide enough
// (M = 5, Dout = 100: 512 of 512 columns), else in a tile of its own.
__host__ __device__ constexpr int dec_bwd_z_cols(int M, int L, int Dout) {
    const int H = 64, z0 = round_up(M * Dout, 16), z1 = L > 1 ? M * H : 0;
    return z0 > z1 ? z0 : z1;
}
__host__ __device__ constexpr bool dec_bwd_z_aliased(int M, int L, int Dout) { return dec_bwd_z_cols(M, L, Dout) <= (M - 1) * 2 * 64; }
__host__ __device__ constexpr size_t dec_bwd_lds_floats(int M, int L, int Dout) {
    const int H = 64, FP = round_up(Dout, 16);
    return (size_t)(M - 1) * kPFloats + (size_t)kDecRows * (M * H + M * 2 * H) + 2 * (size_t)kDecRows * FP + (size_t)kDecRows * (H + 4)
           + (size_t)L * 4 * 2 * 256 + 4 * 4 * kRemTile
           + (dec_bwd_z_aliased(M, L, Dout) ? 0 : (size_t)kDecRows * lds_stride_x(dec_bwd_z_cols(M, L, Dout)));
}

template <int H, int M, int DT>
__global__ __launch_bounds__(256, 1) void dec_bwd_persist_kernel(DecBwdArgs a) {
    static_assert(H == 64, "one column tile per wave");
    constexpr int NKS = 5, ROWS = kDecRows, KAP = M * H, KGP = M * 2 * H, NCT = H / 16, NQ = M * H / 16, DAS = H + 4;
    EEG_DYN_SMEM(sm);
    const int T = a.T, B = a.B, N = a.N, Dout = a.Dout, L = a.L, act = a.act;
    const int FP = round_up(Dout, 16);
    float* Pl = sm;
    float* EC = Pl + (M - 1) * kPFloats;     // [ROWS][KAP]  slot 0 = dC, slots m = P_m^T dC
    float* EG = EC + ROWS * KAP;             // [ROWS][KGP]  slot 0 = [dR|dU], slots m = P_m^T [dR|dU]; cols 2H.. double as Z
    float* DO = EG + ROWS * KGP;             // [ROWS][FP]   total gradient of out_t
    float* DX = DO + ROWS * FP;              // [ROWS][FP]   input gradient of layer 0 at step t (feeds dO_{t-1})
    float* DA = DX + ROWS * FP;              // [ROWS][DAS]  input gradient of layer l > 0 = gradient of h^{l-1}_t
    float* DH = DA + ROWS * DAS;             // [L][4 waves][2][64 lanes] float4: recurrent gradients dh^l, lane-linear
    float* RS0 = DH + L * 4 * 2 * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), lr = lane & 15, lg = lane >> 4;
    float* RS = RS0 + wave * (4 * kRemTile);
    const bool z_alias = dec_bwd_z_aliased(M, L, Dout);
    float* ZT = z_alias ? EG : RS0 + 4 * 4 * kRemTile;           // Z = dXW W^x^T, swizzled like the state tiles
    const int ZS = z_alias ? KGP : lds_stride_x(dec_bwd_z_cols(M, L, Dout)), zc0 = z_alias ? 2 * H : 0;
    const int ct = wave, col = ct * 16 + 4 * lg;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const size_t state = (size_t)B * N * H;
    const int nct_h = H / 16;

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < (int)(ROWS * (KAP + KGP) + 2 * ROWS * FP + ROWS * DAS + L * 4 * 2 * 256); e += 256) EC[e] = 0.f;
        if (!z_alias)
            for (int e = tid; e < ROWS * ZS; e += 256) ZT[e] = 0.f;
        lds_load_polys(Pl, a.P, a.p_batched ? b : 0, M, N);
        __syncthreads();
        float pf[poly_slots<M, NKS>()][NKS];
        load_poly_frags<M, NKS, true>(Pl, pf, lr, lg);
        const int node[2] = {lr, 16 + lr};
        const int rowt[2] = {lr, 16 + (lr & 3)};                       // tile rows (second node tile: 4 rows)
        const bool valid[2] = {lr < N, 16 + lr < N};
        const int nodec[2] = {valid[0] ? lr : N - 1, valid[1] ? 16 + lr : N - 1};
        const int oh[2] = {nodec[0] * H + col, nodec[1] * H + col};
        const int oxw[2] = {node[0] * (3 * H) + col, node[1] * (3 * H) + col};
        float* dhl = DH + (wave * 2) * 256 + 4 * lane;                 // + l * 2048 + nt * 256
        const size_t boff = (size_t)b * N * H;
        // operands of one (layer, step) pair; the next pair's are requested while the current one is processed
        f32x4 nh[2], nr[2], nu[2], nc[2];
        auto fetch = [&](int l, int t) {
            const DecBwdLayerPtrs& lp = a.l[l];
            const size_t so = (size_t)t * state + boff;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                nh[nt] = ld4(lp.hext + so + oh[nt]);                   // hext slot t = h_{t-1}
                nr[nt] = ld4(lp.rs + so + oh[nt]);
                nu[nt] = ld4(lp.us + so + oh[nt]);
                nc[nt] = ld4(lp.cs + so + oh[nt]);
            }
        };
        fetch(L - 1, T - 1);
        for (int t = T - 1; t >= 0; --t) {
            const size_t s = (size_t)t * B + b;
            const bool fb = ((a.feeds_mask >> t) & 1ull) != 0;       // out_t feeds step t+1: its gradient gets DX of that step
            // ---- S0: total gradient of out_t -> DO tile and dOtot
            for (int e = tid; e < N * (Dout / 4); e += 256) {
                const int n = e / (Dout / 4), c4 = e % (Dout / 4);
                f32x4 g = ld4(a.dOut + (s * N + n) * Dout + 4 * c4);
                if (fb) g += ld4(DX + n * FP + 4 * c4);
                st4(DO + n * FP + 4 * c4, g);
                st4(a.dOtot + (s * N + n) * Dout + 4 * c4, g);
            }
            __syncthreads();
            // ---- gradient of h^{L-1}_t through the projection (model.py:188-190): dA = dO W_p
            f32x4 gext[2];
            {
                f32x4 pa[1][2] = {{zero4, zero4}};
                const int wt1[1] = {ct};
                gemm_stream_plain<1, DT, false>(DO, FP, FP, Dout / 4, 1, a.tpack, nct_h, wt1, lane, lr, lg, pa, RS);
                gext[0] = pa[0][0];
                gext[1] = pa[0][1];
            }
            for (int l = L - 1; l >= 0; --l) {
                const DecBwdLayerPtrs& lp = a.l[l];
                const int Fin = l == 0 ? Dout : H;
                float* dxw = lp.dxw + s * N * (3 * H);
                f32x4 hp[2], rr[2], uu[2], cc[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) { hp[nt] = nh[nt]; rr[nt] = nr[nt]; uu[nt] = nu[nt]; cc[nt] = nc[nt]; }
                if (l > 0) fetch(l - 1, t); else if (t > 0) fetch(L - 1, t - 1);
                // ---- E1: blend backward (cell.py:182-210 reversed)
                f32x4 dU[2], dhn[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f32x4 h = hp[nt], u = uu[nt], c = cc[nt];
                    const f32x4 g = valid[nt] ? ld4(dhl + l * 2048 + nt * 256) + gext[nt] : zero4;
                    f32x4 dC, du_;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dc = g[r] * (1.f - u[r]);
                        dC[r] = act == 0 ? dc * (1.f - c[r] * c[r]) : (c[r] > 0.f ? dc : 0.f);
                        du_[r] = g[r] * (h[r] - c[r]) * u[r] * (1.f - u[r]);
                    }
                    if (nt == 0 || lr < 4) st4(EC + lds_sw(rowt[nt], col, KAP), dC);      // zeros on padding nodes
                    if (valid[nt]) {
                        st4(dxw + oxw[nt] + 2 * H, dC);
                        st4(dxw + oxw[nt] + H, du_);
                    }
                    dU[nt] = du_;
                    dhn[nt] = g * u;
                }
                EEG_WAVE_SYNC();
                lds_diffuse_tile<M, NKS, ROWS>(EC, KAP, ct * 16, H, pf, lr, lg);
                __syncthreads();                                         // (1) P_m^T dC complete
                // ---- GEMM1: d(r*h) = [P_m^T dC]_m @ Wc^h^T
                f32x4 acc1[1][2] = {{zero4, zero4}};
                {
                    const int wt1[1] = {ct};
                    gemm_stream_quad<1, NQ, (NQ < 10 ? NQ : 10)>(EC, KAP, lp.b1, NCT, wt1, lane, lr, lg, acc1, RS);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f32x4 drh = acc1[0][nt], rg = rr[nt];          // exact 0 on padding nodes
                    const f32x4 dR = drh * hp[nt] * rg * (1.f - rg);
                    dhn[nt] += drh * rg;
                    if (nt == 0 || lr < 4) {
                        st4(EG + lds_sw(rowt[nt], col, KGP), dR);
                        st4(EG + lds_sw(rowt[nt], H + col, KGP), dU[nt]);
                    }
                    if (valid[nt]) st4(dxw + oxw[nt], dR);
                }
                EEG_WAVE_SYNC();
                lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, ct * 16, 2 * H, pf, lr, lg);
                lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, H + ct * 16, 2 * H, pf, lr, lg);
                __syncthreads();                                         // (2) P_m^T [dR|dU] complete
                // ---- GEMM2: dh = dhn + [P_m^T dG]_m @ Wg^h^T  -> recurrent gradient of this layer for step t-1
                {
                    f32x4 acc2[1][2] = {{dhn[0], dhn[1]}};
                    const int wt1[1] = {ct};
                    gemm_stream_quad<1, 2 * NQ, (2 * NQ < 10 ? 2 * NQ : 10)>(EG, KGP, lp.b2, NCT, wt1, lane, lr, lg, acc2, RS);
                    st4(dhl + l * 2048 + 0 * 256, acc2[0][0]);
                    st4(dhl + l * 2048 + 1 * 256, acc2[0][1]);
                }
                const bool need_dx = l > 0 || (t > 0 && ((a.feeds_mask >> (t - 1)) & 1ull) != 0);
                if (!need_dx) {
                    __syncthreads();                                     // tiles free for the next pair
                    continue;
                }
                __syncthreads();                                         // (3) every wave is done with the hop slots of EG
                // ---- Z = dXW_t @ W^x^T (K = 3H: [dR|dU] from EG slot 0, dC from EC slot 0) -> EG columns 2H.. (M*Fin wide)
                {
                    const int nct_x = round_up(M * Fin, 16) / 16;
                    for (int j0 = wave; j0 < nct_x; j0 += 16) {          // this wave's tiles j0, j0+4, j0+8, j0+12
                        f32x4 z[4][2];
                        int wt4[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            wt4[i] = j0 + 4 * i < nct_x ? j0 + 4 * i : j0;
                            z[i][0] = zero4;
                            z[i][1] = zero4;
                        }
                        gemm_stream_plain<4, 8, true>(EG, KGP, 2 * H, 2 * H / 4, 1, lp.bxt, nct_x, wt4, lane, lr, lg, z, RS);
                        gemm_stream_plain<4, 8, true>(EC, KAP, H, H / 4, 1, lp.bxt + (size_t)(2 * H / 4) * nct_x * 64, nct_x, wt4, lane, lr, lg, z, RS);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (j0 + 4 * i >= nct_x) continue;
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                if (nt == 0 || lr < 4) st4(ZT + lds_sw(rowt[nt], zc0 + wt4[i] * 16 + 4 * lg, ZS), z[i][nt]);
                        }
                    }
                }
                __syncthreads();                                         // (4) Z complete
                // ---- adjoint node mix: dX[n][f] = Z_0[n][f] + sum_{m>=1} sum_q P_m[q][n] Z_m[q][f]
                {
                    const int nctf = round_up(Fin, 16) / 16, nks = ceil_div(N, 4);
                    float* dst = l == 0 ? DX : DA;
                    const int dss = l == 0 ? FP : DAS;
                    for (int tix = wave; tix < 2 * nctf; tix += 4) {
                        const int cti = tix % nctf, rt = tix / nctf;
                        f32x4 acc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = rt * 16 + 4 * lg + r;
                            acc[r] = n < ROWS ? ZT[lds_sw(n, zc0 + cti * 16 + lr, ZS)] : 0.f;
                        }
                        for (int m1 = 0; m1 < M - 1; ++m1) {
                            const float* Pm = Pl + m1 * kPFloats;
                            for (int ks = 0; ks < nks; ++ks) {
                                const int kk = 4 * ks + lg;
                                acc = mfma16(Pm[kk * kPStride + rt * 16 + lr], ZT[lds_sw(kk, zc0 + (m1 + 1) * Fin + cti * 16 + lr, ZS)], acc);
                            }
                        }
                        const int c = cti * 16 + lr;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = rt * 16 + 4 * lg + r;
                            if (n < ROWS && c < Fin) dst[n * dss + c] = n < N ? acc[r] : 0.f;
                        }
                    }
                }
                __syncthreads();                                         // (5) DA / DX complete; tiles free for the next pair
                if (l > 0) {
                    gext[0] = ld4(DA + rowt[0] * DAS + col);
                    gext[1] = ld4(DA + rowt[1] * DAS + col);
                }
            }
        }
        // ---- gradients of the initial states (the encoder's final states)
        if (a.dh0 != nullptr) {
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    if (valid[nt]) st4(a.dh0 + (size_t)l * state + boff + node[nt] * H + col, ld4(dhl + l * 2048 + nt * 256));
        }
    }
}

}  // namespace eeg
