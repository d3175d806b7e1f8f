// BPTT recurrent kernel with STREAMED weights, two clips per CU (64 units, at most 20 nodes).
//
// seq_bwd_kernel keeps the recurrent weights in registers: at M >= 4 hop matrices that is 240+ registers per lane, one
// wave per SIMD, one clip per CU -- and the step is a latency chain that keeps the matrix pipe ≈ half busy (DESIGN.md
// §4.1).  When a launch has more clips than the chip has CUs (cfg5: 512), the kernel below runs TWO workgroups per CU
// instead: no weight lives in registers (the fragments are streamed from L2 a few quads ahead of their MFMAs, as in
// kernels_decoder.h), so a workgroup needs 256 registers and 80 KB of LDS, and the two independent chains of a CU fill
// each other's stalls.  Same operands and outputs as seq_bwd_kernel.  Measured (cfg5, B = 512, M = 5): seq_bwd 2.08 ->
// 1.91 ms.  The forward twin (same construction, gate + candidate GEMMs of 2 + 1 tiles) was 6 % SLOWER than the
// register-resident seq_fwd_kernel and is not kept (2.06 -> 2.18 ms; with buffer-descriptor operand accesses and no
// spill 2.13 ms at 3 / 4 quads of gate / candidate weights in flight, 2.17 ms at 5 / 8).
#pragma once
#include "kernels_decoder.h"

namespace eeg {

// (the hop polynomials are only staged through the tile area: after load_poly_frags they live in registers -- at M = 5
//  the tiles of two workgroups take 150 of the 160 KB of a CU)
__host__ __device__ constexpr size_t seq_stream_bwd_lds_floats(int M) {
    const size_t tiles = (size_t)kDecRows * (M * 64 + M * 128), polys = (size_t)(M - 1) * kPFloats;
    return tiles > polys ? tiles : polys;
}

// BPTT, same contract as seq_bwd_kernel (operands one step ahead, d_at_end / d_at_len / lengths, dXW, dh0 and the
// per-clip bias-gradient sums), weights b1 / b2 streamed.
template <int H, int M>
__global__ __launch_bounds__(256, 2) void seq_bwd_stream_kernel(
    const float* __restrict__ Hseq, const float* __restrict__ h0, const float* __restrict__ Rs,
    const float* __restrict__ Us, const float* __restrict__ Cs, const float* __restrict__ dHseq,
    const float* __restrict__ d_at_end, const float* __restrict__ d_at_len, const long long* __restrict__ lengths,
    const float* __restrict__ P, int p_batched, const float* __restrict__ b1p, const float* __restrict__ b2p,
    float* __restrict__ dXW, float* __restrict__ dh0, float* __restrict__ dbias_part, int T, int B, int N, int act) {
    static_assert(H == 64, "one column tile per wave");
    constexpr int NKS = 5, ROWS = kDecRows, KAP = M * H, KGP = M * 2 * H, NCT = H / 16, NQ = M * H / 16;
    constexpr int PD = NQ < 3 ? NQ : 3;           // quads of weights in flight: deeper costs registers, and spills are costly here (2: +1 %, 4: +1 %, 6: +12 %)
    EEG_DYN_SMEM(sm);
    constexpr int TILES = ROWS * (KAP + KGP);
    float* Pl = sm;                         // staging only (aliases the tiles)
    float* EC = sm;                         // [ROWS][KAP]  slot 0 = dC, slots m = P_m^T dC
    float* EG = EC + ROWS * KAP;            // [ROWS][KGP]  slot 0 = [dR|dU], slots m = P_m^T [dR|dU]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), lr = lane & 15, lg = lane >> 4;
    const int ct = wave, col = ct * 16 + 4 * lg;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int wt1[1] = {ct};
    const wbuf_t bH = make_wbuf(Hseq), bH0 = make_wbuf(h0 != nullptr ? h0 : Hseq), bR = make_wbuf(Rs), bU = make_wbuf(Us),
                 bC = make_wbuf(Cs), bG = make_wbuf(dHseq != nullptr ? dHseq : Hseq), bX = make_wbuf(dXW);

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();                                                // previous clip: the bias reduction has read EC
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
        __syncthreads();                                                // polynomials are in registers: the area becomes the tiles
        for (int e = tid; e < TILES; e += 256) EC[e] = 0.f;
        __syncthreads();
        const int node[2] = {lr, 16 + lr};
        const bool valid[2] = {lr < N, 16 + lr < N};
        const int nodec[2] = {valid[0] ? lr : N - 1, valid[1] ? 16 + lr : N - 1};
        const int oh[2] = {nodec[0] * H + col, nodec[1] * H + col};
        const int oxw[2] = {node[0] * (3 * H) + col, node[1] * (3 * H) + col};
        // remainder nodes 16..19: one element per lane (mfma_nodes32 L1) -- lane (lr, lg) <-> node 16 + lg, column ct*16 + lr;
        // every nt = 1 quantity below lives in component 0 of its vector
        const int node1 = 16 + lg, col1 = ct * 16 + lr;
        const bool valid1 = node1 < N;
        const int oh1 = (valid1 ? node1 : N - 1) * H + col1, oxw1 = node1 * (3 * H) + col1;
        const int lc1 = lds_sw(node1, col1, KAP), lg1 = lds_sw(node1, col1, KGP), lu1 = lds_sw(node1, H + col1, KGP);
        f32x4 dh[2] = {zero4, zero4}, sb_r = zero4, sb_u = zero4, sb_c = zero4;
        float s1_r = 0.f, s1_u = 0.f, s1_c = 0.f;
        const size_t tstride = (size_t)B * N * H, boff = (size_t)b * N * H;
        // operands through buffer descriptors: one per-lane VGPR offset per node tile, the step offset in an SGPR
        // (64-bit per-lane addresses of six arrays would not fit next to the weight stream in 256 registers)
        f32x4 nh[2], nr[2], nu[2], nc[2], ng[2];
        auto fetch = [&](int t) {
            const unsigned so = (unsigned)((size_t)t * tstride + boff);
            {
                const unsigned o = oh[0];
                nh[0] = t > 0 ? wbuf_ld4(bH, o, so - (unsigned)tstride) : (h0 != nullptr ? wbuf_ld4(bH0, o, (unsigned)boff) : zero4);
                nr[0] = wbuf_ld4(bR, o, so);
                nu[0] = wbuf_ld4(bU, o, so);
                nc[0] = wbuf_ld4(bC, o, so);
                f32x4 g = dHseq != nullptr ? wbuf_ld4(bG, o, so) : zero4;
                if (d_at_end != nullptr && t == T - 1) g += ld4(d_at_end + boff + o);
                if (t == t_len) g += ld4(d_at_len + boff + o);
                ng[0] = g;
            }
            {
                const unsigned o = oh1;
                nh[1] = (f32x4){t > 0 ? wbuf_ld(bH, o, so - (unsigned)tstride) : (h0 != nullptr ? wbuf_ld(bH0, o, (unsigned)boff) : 0.f), 0.f, 0.f, 0.f};
                nr[1] = (f32x4){wbuf_ld(bR, o, so), 0.f, 0.f, 0.f};
                nu[1] = (f32x4){wbuf_ld(bU, o, so), 0.f, 0.f, 0.f};
                nc[1] = (f32x4){wbuf_ld(bC, o, so), 0.f, 0.f, 0.f};
                float g1 = dHseq != nullptr ? wbuf_ld(bG, o, so) : 0.f;
                if (d_at_end != nullptr && t == T - 1) g1 += d_at_end[boff + o];
                if (t == t_len) g1 += d_at_len[boff + o];
                ng[1] = (f32x4){g1, 0.f, 0.f, 0.f};
            }
        };
        fetch(T - 1);
        float wq1[PD + 1][4][1], wq2[PD + 1][4][1];
        quad_prefetch<1, NQ, PD>(b1p, NCT, wt1, lane, wq1);
        for (int t = T - 1; t >= 0; --t) {
            const unsigned sx = (unsigned)(((size_t)t * B + b) * N * (3 * H));
            f32x4 hp[2], rr[2], uu[2], cc[2], gg[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) { hp[nt] = nh[nt]; rr[nt] = nr[nt]; uu[nt] = nu[nt]; cc[nt] = nc[nt]; gg[nt] = ng[nt]; }
            if (t > 0) fetch(t - 1);
            // ---- E1: gate blend backward on the owned elements (padding nodes zeroed)
            f32x4 dU[2], dhn[2];
            {
                const f32x4 h = hp[0], u = uu[0], c = cc[0];
                const f32x4 g = valid[0] ? dh[0] + gg[0] : zero4;
                f32x4 dC, du_;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dc = g[r] * (1.f - u[r]);
                    dC[r] = act == 0 ? dc * (1.f - c[r] * c[r]) : (c[r] > 0.f ? dc : 0.f);
                    du_[r] = g[r] * (h[r] - c[r]) * u[r] * (1.f - u[r]);
                }
                st4(EC + lds_sw(lr, col, KAP), dC);
                if (valid[0]) {
                    wbuf_st4(bX, oxw[0] + 2 * H, sx, dC);
                    wbuf_st4(bX, oxw[0] + H, sx, du_);
                }
                sb_c += dC;
                sb_u += du_;
                dU[0] = du_;
                dhn[0] = g * u;
                const float h1 = hp[1][0], u1 = uu[1][0], c1 = cc[1][0];
                const float g1 = valid1 ? dh[1][0] + gg[1][0] : 0.f;
                const float dc1 = g1 * (1.f - u1);
                const float dC1 = act == 0 ? dc1 * (1.f - c1 * c1) : (c1 > 0.f ? dc1 : 0.f);
                const float du1 = g1 * (h1 - c1) * u1 * (1.f - u1);
                EC[lc1] = dC1;
                if (valid1) {
                    wbuf_st1(bX, oxw1 + 2 * H, sx, dC1);
                    wbuf_st1(bX, oxw1 + H, sx, du1);
                }
                s1_c += dC1;
                s1_u += du1;
                dU[1] = (f32x4){du1, 0.f, 0.f, 0.f};
                dhn[1] = (f32x4){g1 * u1, 0.f, 0.f, 0.f};
            }
            EEG_WAVE_SYNC();
            lds_diffuse_tile<M, NKS, ROWS>(EC, KAP, ct * 16, H, pf, lr, lg);
            __syncthreads();                                             // (1) P_m^T dC complete
            // ---- GEMM1: d(r*h) = [P_m^T dC]_m @ Wc^h^T
            f32x4 acc[1][2] = {{zero4, zero4}};
            gemm_stream_quad<1, NQ, PD, true>(EC, KAP, b1p, NCT, wt1, lane, lr, lg, acc, wq1);
            quad_prefetch<1, 2 * NQ, PD>(b2p, NCT, wt1, lane, wq2);
            {
                const f32x4 drh = acc[0][0], rg = rr[0];                 // exact 0 on padding nodes
                const f32x4 dR = drh * hp[0] * rg * (1.f - rg);
                dhn[0] += drh * rg;
                st4(EG + lds_sw(lr, col, KGP), dR);
                st4(EG + lds_sw(lr, H + col, KGP), dU[0]);
                if (valid[0]) wbuf_st4(bX, oxw[0], sx, dR);
                sb_r += dR;
                acc[0][0] = dhn[0];
                const float drh1 = acc[0][1][0], rg1 = rr[1][0];
                const float dR1 = drh1 * hp[1][0] * rg1 * (1.f - rg1);
                dhn[1][0] += drh1 * rg1;
                EG[lg1] = dR1;
                EG[lu1] = dU[1][0];
                if (valid1) wbuf_st1(bX, oxw1, sx, dR1);
                s1_r += dR1;
                acc[0][1] = (f32x4){dhn[1][0], 0.f, 0.f, 0.f};
            }
            EEG_WAVE_SYNC();
            lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, ct * 16, 2 * H, pf, lr, lg);
            lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, H + ct * 16, 2 * H, pf, lr, lg);
            __syncthreads();                                             // (2) P_m^T [dR|dU] complete
            // ---- GEMM2: dh = dhn + [P_m^T dG]_m @ Wg^h^T
            gemm_stream_quad<1, 2 * NQ, PD, true>(EG, KGP, b2p, NCT, wt1, lane, lr, lg, acc, wq2);
            if (t > 0) quad_prefetch<1, NQ, PD>(b1p, NCT, wt1, lane, wq1);
            dh[0] = acc[0][0];
            dh[1] = acc[0][1];
        }
        // ---- epilogue: dh0 and the per-clip bias-gradient partial sums (fixed-order node reduction)
        __syncthreads();                                                // all waves done with the tiles
        float* red = EC;                                                // [3H][16] + [3H][4] (the remainder elements)
        float* red1 = EC + 3 * H * 16;
        if (dh0 != nullptr) {
            if (valid[0]) st4(dh0 + boff + lr * H + col, dh[0]);
            if (valid1) dh0[boff + node1 * H + col1] = dh[1][0];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[(0 * H + col + r) * 16 + lr] = sb_r[r];
            red[(1 * H + col + r) * 16 + lr] = sb_u[r];
            red[(2 * H + col + r) * 16 + lr] = sb_c[r];
        }
        red1[(0 * H + col1) * 4 + lg] = s1_r;
        red1[(1 * H + col1) * 4 + lg] = s1_u;
        red1[(2 * H + col1) * 4 + lg] = s1_c;
        __syncthreads();
        for (int j = tid; j < 3 * H; j += 256) {
            float sacc = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) sacc += red[j * 16 + q];
            sacc += (red1[j * 4] + red1[j * 4 + 1]) + (red1[j * 4 + 2] + red1[j * 4 + 3]);
            dbias_part[(size_t)b * 3 * H + j] = sacc;
        }
    }
}

}  // namespace eeg
