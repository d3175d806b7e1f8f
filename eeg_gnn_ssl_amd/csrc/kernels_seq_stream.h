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
//  the tiles and the hand-over scratch of two workgroups fill the 160 KB of a CU exactly)
__host__ __device__ constexpr size_t seq_stream_bwd_lds_floats(int M) {
    const size_t tiles = (size_t)kDecRows * (M * 64 + M * 128), polys = (size_t)(M - 1) * kPFloats;
    return (tiles > polys ? tiles : polys) + 4 * kRemTile;
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
    constexpr int TILES = ROWS * (KAP + KGP), POLYS = (M - 1) * kPFloats;
    float* Pl = sm;                         // staging only (aliases the tiles)
    float* EC = sm;                         // [ROWS][KAP]  slot 0 = dC, slots m = P_m^T dC
    float* EG = EC + ROWS * KAP;            // [ROWS][KGP]  slot 0 = [dR|dU], slots m = P_m^T [dR|dU]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), lr = lane & 15, lg = lane >> 4;
    float* RS = sm + (TILES > POLYS ? TILES : POLYS) + wave * kRemTile;
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
        const int rowt[2] = {lr, 16 + (lr & 3)};
        const bool valid[2] = {lr < N, 16 + lr < N};
        const int nodec[2] = {valid[0] ? lr : N - 1, valid[1] ? 16 + lr : N - 1};
        const int oh[2] = {nodec[0] * H + col, nodec[1] * H + col};
        const int oxw[2] = {node[0] * (3 * H) + col, node[1] * (3 * H) + col};
        f32x4 dh[2] = {zero4, zero4}, sb_r = zero4, sb_u = zero4, sb_c = zero4;
        const size_t tstride = (size_t)B * N * H, boff = (size_t)b * N * H;
        // operands through buffer descriptors: one per-lane VGPR offset per node tile, the step offset in an SGPR
        // (64-bit per-lane addresses of six arrays would not fit next to the weight stream in 256 registers)
        f32x4 nh[2], nr[2], nu[2], nc[2], ng[2];
        auto fetch = [&](int t) {
            const unsigned so = (unsigned)((size_t)t * tstride + boff);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const unsigned o = oh[nt];
                nh[nt] = t > 0 ? wbuf_ld4(bH, o, so - (unsigned)tstride) : (h0 != nullptr ? wbuf_ld4(bH0, o, (unsigned)boff) : zero4);
                nr[nt] = wbuf_ld4(bR, o, so);
                nu[nt] = wbuf_ld4(bU, o, so);
                nc[nt] = wbuf_ld4(bC, o, so);
                f32x4 g = dHseq != nullptr ? wbuf_ld4(bG, o, so) : zero4;
                if (d_at_end != nullptr && t == T - 1) g += ld4(d_at_end + boff + o);
                if (t == t_len) g += ld4(d_at_len + boff + o);
                ng[nt] = g;
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
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 h = hp[nt], u = uu[nt], c = cc[nt];
                const f32x4 g = valid[nt] ? dh[nt] + gg[nt] : zero4;
                f32x4 dC, du_;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dc = g[r] * (1.f - u[r]);
                    dC[r] = act == 0 ? dc * (1.f - c[r] * c[r]) : (c[r] > 0.f ? dc : 0.f);
                    du_[r] = g[r] * (h[r] - c[r]) * u[r] * (1.f - u[r]);
                }
                if (nt == 0 || lr < 4) st4(EC + lds_sw(rowt[nt], col, KAP), dC);
                if (valid[nt]) {
                    wbuf_st4(bX, oxw[nt] + 2 * H, sx, dC);
                    wbuf_st4(bX, oxw[nt] + H, sx, du_);
                }
                sb_c += dC;
                sb_u += du_;
                dU[nt] = du_;
                dhn[nt] = g * u;
            }
            EEG_WAVE_SYNC();
            lds_diffuse_tile<M, NKS, ROWS>(EC, KAP, ct * 16, H, pf, lr, lg);
            __syncthreads();                                             // (1) P_m^T dC complete
            // ---- GEMM1: d(r*h) = [P_m^T dC]_m @ Wc^h^T
            f32x4 acc[1][2] = {{zero4, zero4}};
            gemm_stream_quad<1, NQ, PD, true>(EC, KAP, b1p, NCT, wt1, lane, lr, lg, acc, RS, wq1);
            quad_prefetch<1, 2 * NQ, PD>(b2p, NCT, wt1, lane, wq2);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 drh = acc[0][nt], rg = rr[nt];               // exact 0 on padding nodes
                const f32x4 dR = drh * hp[nt] * rg * (1.f - rg);
                dhn[nt] += drh * rg;
                if (nt == 0 || lr < 4) {
                    st4(EG + lds_sw(rowt[nt], col, KGP), dR);
                    st4(EG + lds_sw(rowt[nt], H + col, KGP), dU[nt]);
                }
                if (valid[nt]) wbuf_st4(bX, oxw[nt], sx, dR);
                sb_r += dR;
                acc[0][nt] = dhn[nt];
            }
            EEG_WAVE_SYNC();
            lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, ct * 16, 2 * H, pf, lr, lg);
            lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, H + ct * 16, 2 * H, pf, lr, lg);
            __syncthreads();                                             // (2) P_m^T [dR|dU] complete
            // ---- GEMM2: dh = dhn + [P_m^T dG]_m @ Wg^h^T
            gemm_stream_quad<1, 2 * NQ, PD, true>(EG, KGP, b2p, NCT, wt1, lane, lr, lg, acc, RS, wq2);
            if (t > 0) quad_prefetch<1, NQ, PD>(b1p, NCT, wt1, lane, wq1);
            dh[0] = acc[0][0];
            dh[1] = acc[0][1];
        }
        // ---- epilogue: dh0 and the per-clip bias-gradient partial sums (fixed-order node reduction)
        __syncthreads();                                                // all waves done with the tiles
        float* red = EC;                                                // [3H][16]
        if (dh0 != nullptr) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                if (valid[nt]) st4(dh0 + boff + node[nt] * H + col, dh[nt]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[(0 * H + col + r) * 16 + lr] = sb_r[r];
            red[(1 * H + col + r) * 16 + lr] = sb_u[r];
            red[(2 * H + col + r) * 16 + lr] = sb_c[r];
        }
        __syncthreads();
        for (int j = tid; j < 3 * H; j += 256) {
            float sacc = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) sacc += red[j * 16 + q];
            dbias_part[(size_t)b * 3 * H + j] = sacc;
        }
    }
}

}  // namespace eeg
