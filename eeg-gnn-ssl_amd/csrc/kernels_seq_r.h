// Persistent DCGRU sequence kernels, wave-specialised "16 + remainder" variant for the 19-electrode
// EEG montage (16 < N <= 20 nodes, M <= 3 hop matrices).  Same algorithm, layouts and interfaces
// as kernels_seq.h; the difference is how the node dimension is mapped onto a CU:
//
//   * kernels_seq.h pads 19 nodes to two 16-node MFMA tiles and burns 41 % of its matrix-pipe
//     issue slots on padding rows, with one wave per SIMD (no latency hiding in the epilogues).
//   * here a workgroup has 8 waves = 2 per SIMD:
//       - TILE waves 0..3: nodes 0..15 as ONE MFMA tile (transposed issue, 16-byte epilogues);
//       - REM  waves 4..7: the NR = N-16 remainder nodes with plain VALU FMAs, from their own
//         register copy of the same weight fragments: lane (c = lane&15, g = lane>>4) holds
//         W[k in its quad-permuted k subset][col c], accumulates the partial dot products of its k
//         subset for the NR nodes, and the four lane groups are summed with butterfly shuffles.
//     The matrix pipe and the VALU pipe of a SIMD run concurrently, so the remainder work and
//     the other wave's epilogue hide under the MFMA stream.  (VALU instructions cannot source
//     AGPRs, which is why the two roles are separate waves: each fits the 256-register budget.)
//
// Wave w (tile) and w+4 (rem) own column tile ct = w + 4*i of r, u, c, h (and dR/dU/dC).
//   tile wave: lane owns node n = lane&15, columns 16*ct + 4*(lane>>4) .. +3   (16-byte ops)
//   rem wave : lanes with (lane>>4) < NR own node 16 + (lane>>4), column 16*ct + (lane&15)
// Four workgroup barriers per step: tile and rem waves exchange rows through the LDS tiles.
#pragma once
#include "common.h"
#include "kernels_seq.h"

#ifndef EEG_REM_PRIO
#define EEG_REM_PRIO 2
#endif
namespace eeg {

constexpr int kRNKS = 5;      // node-mix k-steps: nodes 0..19
constexpr int kRRows = 20;    // LDS tile rows (row 19 is a zero pad row)

template <int H, int M>
struct SeqGeomR {
    static constexpr int KA = M * H, KAP = lds_stride_q(KA), KS = KA / 4;
    static constexpr int KG = M * 2 * H, KGP = lds_stride_q(KG), KSG = KG / 4;
    static constexpr int NGT = 2 * H / 16, NCT = H / 16, CT = ceil_div(NCT, 4);
    static constexpr size_t fwd_lds_floats() { return (size_t)(M - 1) * kPFloats + 2 * kRRows * KAP; }
    static constexpr size_t bwd_lds_floats() {
        const size_t tiles = (size_t)kRRows * KAP + (size_t)kRRows * KGP, red = (size_t)3 * H * 20;
        return (size_t)(M - 1) * kPFloats + (tiles > red ? tiles : red);
    }
};

// Sum over the 4 lane groups (lanes l, l^16, l^32, l^48); every lane gets the total.
// gfx950: v_permlane16_swap(x, x) leaves [x0 x0 x2 x2] / [x1 x1 x3 x3] (rows of 16 lanes), so the
// sum of the two results is the xor-16 butterfly; v_permlane32_swap likewise for xor-32: four VALU
// instructions, no LDS crossbar round trips (ds_bpermute) on the latency-critical remainder path.
__device__ __forceinline__ float lg_allreduce(float v) {
#if defined(EEG_SIMT_EMU)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
#else
    const unsigned x = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const float s1 = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    const unsigned y = __builtin_bit_cast(unsigned, s1);
    auto b = __builtin_amdgcn_permlane32_swap(y, y, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
#endif
}
template <int NR>
__device__ __forceinline__ float pick(const float (&v)[NR], int lg) {   // v[lg] without dynamic indexing
    float r = v[0];
#pragma unroll
    for (int j = 1; j < NR; ++j) r = lg == j ? v[j] : r;
    return r;
}

// ---- tile-wave helpers ------------------------------------------------------------------------
// polynomial fragments of the 16-node tile (B operand of the transposed node mix)
template <int M, bool ADJ>
__device__ __forceinline__ void load_poly_tile(const float* Pl, float (&pf0)[M - 1][kRNKS], int lr, int lg) {
#pragma unroll
    for (int m1 = 0; m1 < M - 1; ++m1)
#pragma unroll
        for (int ks = 0; ks < kRNKS; ++ks) {
            const int q = 4 * ks + lg;
            pf0[m1][ks] = ADJ ? Pl[m1 * kPFloats + q * kPStride + lr] : Pl[m1 * kPFloats + lr * kPStride + q];
        }
}
// rows 0..15 of the hop slots m = 1..M-1 of one 16-column tile
template <int M>
__device__ __forceinline__ void diffuse_tile16(float* buf, int stride, int src_col, int slot_w,
                                               const float (&pf0)[M - 1][kRNKS], int lr, int lg) {
    float b[kRNKS];
#pragma unroll
    for (int ks = 0; ks < kRNKS; ++ks) b[ks] = buf[(4 * ks + lg) * stride + src_col + lr];
    f32x4 acc[M - 1];
#pragma unroll
    for (int m1 = 0; m1 < M - 1; ++m1) acc[m1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < kRNKS; ++ks)
#pragma unroll
        for (int m1 = 0; m1 < M - 1; ++m1) acc[m1] = mfma16(b[ks], pf0[m1][ks], acc[m1]);
#pragma unroll
    for (int m1 = 0; m1 < M - 1; ++m1) st4(buf + lr * stride + (m1 + 1) * slot_w + src_col + 4 * lg, acc[m1]);
}
// acc[i] += W-frag[i] x X(nodes 0..15)^T; two accumulator chains per tile when NT == 1
template <int NT, int NKS>
__device__ __forceinline__ void mfma_nodes16(const float* __restrict__ X, int stride, int lr, int lg,
                                             const float (&w)[NT][NKS], f32x4 (&acc)[NT]) {
    static_assert(NKS % 4 == 0, "K must be a multiple of 16");
    const float* p0 = X + lr * stride + 4 * lg;
    f32x4 acc2[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc2[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a0 = ld4(p0);
#pragma unroll
    for (int q = 0; q < NKS / 4; ++q) {
        f32x4 n0 = a0;
        if (q + 1 < NKS / 4) n0 = ld4(p0 + 16 * (q + 1));
        EEG_SCHED_FENCE();
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (NT == 1 && (j4 & 1)) acc2[i] = mfma16(w[i][4 * q + j4], a0[j4], acc2[i]);
                else acc[i] = mfma16(w[i][4 * q + j4], a0[j4], acc[i]);
            }
        EEG_SCHED_FENCE();
        a0 = n0;
    }
    if (NT == 1) {
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] += acc2[i];
    }
}

// ---- rem-wave helpers -------------------------------------------------------------------------
// rows 16..16+NR-1 of the hop slots of one 16-column tile.  The rem waves are far from the critical
// path but share the kernel's 256-register budget with the tile waves, so their polynomial
// coefficients stay in LDS (Pl, zero padded 32x32 per hop) instead of registers.
template <int M, int NR, bool ADJ>
__device__ __forceinline__ void diffuse_rem(float* buf, int stride, int src_col, int slot_w,
                                            const float* __restrict__ Pl, int lr, int lg) {
    float b[kRNKS], v[M - 1][NR];
#pragma unroll
    for (int ks = 0; ks < kRNKS; ++ks) b[ks] = buf[(4 * ks + lg) * stride + src_col + lr];
    // all partial sums first (independent), then all butterflies, then the stores: nothing in
    // between aliases the LDS tile, so the reads / shuffles pipeline instead of serialising
#pragma unroll
    for (int m1 = 0; m1 < M - 1; ++m1)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            float a = 0.f;
#pragma unroll
            for (int ks = 0; ks < kRNKS; ++ks) {
                const int q = 4 * ks + lg;
                const float p = ADJ ? Pl[m1 * kPFloats + q * kPStride + 16 + j] : Pl[m1 * kPFloats + (16 + j) * kPStride + q];
                a = fmaf(p, b[ks], a);
            }
            v[m1][j] = a;
        }
#pragma unroll
    for (int m1 = 0; m1 < M - 1; ++m1)
#pragma unroll
        for (int j = 0; j < NR; ++j) v[m1][j] = lg_allreduce(v[m1][j]);
    if (lg < NR) {
#pragma unroll
        for (int m1 = 0; m1 < M - 1; ++m1) buf[(16 + lg) * stride + (m1 + 1) * slot_w + src_col + lr] = pick<NR>(v[m1], lg);
    }
}
// out[i] = (sum over all k) W[k][col lr of tile i] * X[node 16+lg][k]  for the lane's own node.
// The remainder GEMM runs on the matrix pipe too, as v_mfma_f32_4x4x1 (16 independent 4x4 outer
// products per instruction, a quarter of the cost of a 16x16x4): block = (lane group lg, column
// quad), so the B operand is the SAME weight register as in the 16-node path (lane (lr, lg) holds
// W[k(ks, lg)][col lr]) and the A operand is X[node 16 + (lane & 3)][k(ks, lg)] (row 19 is the zero
// pad row).  Result register r of a lane = this lane group's partial of out[node 16 + r][col lr];
// the four lane groups are summed with the permlane butterfly.  (VALU FMAs were tried first: fp32
// MFMA and VALU share the FP32 ALUs, so that stream only ran in the gaps of the tile wave's MFMAs.)
template <int NT, int NKS>
__device__ __forceinline__ void mfma_nodes_rem(const float* __restrict__ X, int stride, int lane, int lg,
                                               const float (&w)[NT][NKS], f32x4 (&acc)[NT]) {
    static_assert(NKS % 4 == 0, "K must be a multiple of 16");
    const float* p = X + (16 + (lane & 3)) * stride + 4 * lg;
    f32x4 acc2[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // a quad of k-steps is only 4*NT short MFMAs (8 cycles each): keep PD quads of operands in flight
    constexpr int NQ = NKS / 4, PD = NQ < 4 ? NQ : 4;
    f32x4 ring[PD];
#pragma unroll
    for (int d = 0; d < PD; ++d) ring[d] = ld4(p + 16 * d);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const f32x4 a0 = ring[q % PD];
        if (q + PD < NQ) ring[q % PD] = ld4(p + 16 * (q + PD));
        EEG_SCHED_FENCE();
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (j4 & 1) acc2[i] = mfma4(a0[j4], w[i][4 * q + j4], acc2[i]);
                else acc[i] = mfma4(a0[j4], w[i][4 * q + j4], acc[i]);
            }
        EEG_SCHED_FENCE();
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] += acc2[i];
}
// sum the lane-group partials of mfma_nodes_rem; out[i] = the value of the lane's own node 16 + lg
template <int NT, int NR>
__device__ __forceinline__ void reduce_rem(const f32x4 (&acc)[NT], int lg, float (&out)[NT]) {
    static_assert(NR <= 4, "at most 4 remainder nodes");
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        float tot[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) tot[j] = lg_allreduce(acc[i][j]);
        out[i] = pick<NR>(tot, lg);
    }
}

// ================================================================================================
template <int H, int M, int NR, bool PROBE = false>
__global__ __launch_bounds__(512, 2) void seq_fwd_r_kernel(
    const float* __restrict__ XW, const float* __restrict__ h0, const float* __restrict__ P, int p_batched,
    const float* __restrict__ bhg, const float* __restrict__ bhc,
    float* __restrict__ Hseq, float* __restrict__ Rs, float* __restrict__ Us, float* __restrict__ Cs,
    float* __restrict__ RHs, float* __restrict__ Hpl, float* __restrict__ RHpl, size_t plane_stride,
    int T, int B, int N, int act, long long* probe) {
    using G = SeqGeomR<H, M>;
    constexpr int KAP = G::KAP, KS = G::KS, CT = G::CT, NGT = G::NGT, NCT = G::NCT;
    EEG_DYN_SMEM(sm);
    float* Pl = sm;
    float* A = Pl + (M - 1) * kPFloats;     // [20][KAP]  slot 0 = h, slots m = P_m h
    float* A2 = A + kRRows * KAP;           // [20][KAP]  slot 0 = r*h
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const bool is_tile = wave8 < 4;
    const int wave = wave8 & 3;
    const int b = blockIdx.x;
    const bool save = Rs != nullptr;
    const bool remv = lg < NR;
    // The hop rows of step t (slots 1..M-1 of a complete LDS tile) are kept in global memory as a
    // by-product for the hoisted weight-gradient GEMMs.  The REM waves copy them out (they have slack;
    // the TILE waves are the critical path): 256 lanes x 16 bytes per pass.
    auto copy_planes = [&](const float* tile, float* planes, int t) {
        if (planes == nullptr) return;
        float* g = planes + ((size_t)t * B + b) * N * H;
        constexpr int Q = H / 4;
        for (int e = wave * 64 + lane; e < N * (M - 1) * Q; e += 256) {
            const int node = e / ((M - 1) * Q), r = e % ((M - 1) * Q), m1 = r / Q, c4 = r % Q;
            st4(g + (size_t)m1 * plane_stride + node * H + 4 * c4, ld4(tile + node * KAP + (m1 + 1) * H + 4 * c4));
        }
    };

    // both roles keep the gate/candidate fragments of their column tile(s) in registers
    float wg[2 * CT][KS], wc[CT][KS];       // [i] = r tile, [CT + i] = u tile
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
    for (int e = tid; e < 2 * kRRows * KAP; e += 512) A[e] = 0.f;
    lds_load_polys(Pl, P, p_batched ? b : 0, M, N);
    __syncthreads();
    if (h0 != nullptr) {
        for (int e = tid; e < N * H; e += 512) A[(e / H) * KAP + (e % H)] = h0[(size_t)b * N * H + e];
    } else {      // zero initial state: clear this clip's row of the slot in front of Hseq (= Hext slot 0, read by the backward)
        for (int e = tid; e < N * H; e += 512) (Hseq - (size_t)B * N * H)[(size_t)b * N * H + e] = 0.f;
    }
    bool own[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) own[i] = wave + 4 * i < NCT;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    if (is_tile) {
        // ===================================== TILE waves =====================================
        PhaseProbe<PROBE> pp;
        pp.start();
        float pf0[M - 1][kRNKS];
        load_poly_tile<M, false>(Pl, pf0, lr, lg);
        int oxw[CT], oh[CT], lt[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ct = own[i] ? wave + 4 * i : 0;
            oxw[i] = lr * (3 * H) + ct * 16 + 4 * lg;
            oh[i] = lr * H + ct * 16 + 4 * lg;
            lt[i] = lr * KAP + ct * 16 + 4 * lg;
        }
#pragma unroll
        for (int i = 0; i < CT; ++i)
            if (own[i]) diffuse_tile16<M>(A, KAP, (wave + 4 * i) * 16, H, pf0, lr, lg);
        // XW of step t+1 is fetched in the middle of step t, AHEAD of that step's h / c stores in the
        // memory queue: waiting for it later does not have to drain those stores
        f32x4 nr[CT], nu[CT], nc[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const float* xw0 = XW + (size_t)b * N * (3 * H);
            nr[i] = ld4(xw0 + oxw[i]); nu[i] = ld4(xw0 + oxw[i] + H); nc[i] = ld4(xw0 + oxw[i] + 2 * H);
        }
        for (int t = 0; t < T; ++t) {
            const size_t s = (size_t)t * B + b;
            f32x4 xr[CT], xu[CT], xc[CT], ag[2 * CT], ac[CT], ug[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                ag[i] = zero4; ag[CT + i] = zero4; ac[i] = zero4;
                xr[i] = nr[i]; xu[i] = nu[i]; xc[i] = nc[i];
            }
            __syncthreads();                                        // (1) hops(h) complete
            pp.mark(0);
            mfma_nodes16<2 * CT, KS>(A, KAP, lr, lg, wg, ag);
            pp.mark(1);
            float* r_t = Rs + s * N * H;
            float* rh_t = RHs + s * N * H;
            float* u_t = Us + s * N * H;
#pragma unroll
            for (int i = 0; i < CT; ++i)
                if (own[i]) {
                    f32x4 rg, u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        rg[r] = sigmoidf_(ag[i][r] + xr[i][r]);
                        u[r] = sigmoidf_(ag[CT + i][r] + xu[i][r]);
                    }
                    ug[i] = u;
                    const f32x4 rh = rg * ld4(A + lt[i]);
                    st4(A2 + lt[i], rh);
                    if (save) {
                        st4(r_t + oh[i], rg);
                        st4(rh_t + oh[i], rh);
                        st4(u_t + oh[i], u);
                    }
                }
            __syncthreads();                                        // (1b) r*h complete (all nodes)
            pp.mark(2);
#pragma unroll
            for (int i = 0; i < CT; ++i)
                if (own[i]) diffuse_tile16<M>(A2, KAP, (wave + 4 * i) * 16, H, pf0, lr, lg);
            __syncthreads();                                        // (2) hops(r*h) complete
            pp.mark(3);
            if (t + 1 < T) {
                const float* xwn = XW + (s + B) * N * (3 * H);
#pragma unroll
                for (int i = 0; i < CT; ++i) {
                    nr[i] = ld4(xwn + oxw[i]); nu[i] = ld4(xwn + oxw[i] + H); nc[i] = ld4(xwn + oxw[i] + 2 * H);
                }
            }
            mfma_nodes16<CT, KS>(A2, KAP, lr, lg, wc, ac);
            pp.mark(4);
            float* h_t = Hseq + s * N * H;
            float* c_t = Cs + s * N * H;
#pragma unroll
            for (int i = 0; i < CT; ++i)
                if (own[i]) {
                    const f32x4 u = ug[i], h = ld4(A + lt[i]);
                    f32x4 c, hn;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pre = ac[i][r] + xc[i][r];
                        c[r] = act == 0 ? tanhf_(pre) : fmaxf(pre, 0.f);
                        hn[r] = u[r] * h[r] + (1.f - u[r]) * c[r];
                    }
                    st4(A + lt[i], hn);
                    st4(h_t + oh[i], hn);
                    if (save) st4(c_t + oh[i], c);
                }
            __syncthreads();                                        // (2b) h_t complete (all nodes)
            pp.mark(5);
            if (t + 1 < T) {
#pragma unroll
                for (int i = 0; i < CT; ++i)
                    if (own[i]) diffuse_tile16<M>(A, KAP, (wave + 4 * i) * 16, H, pf0, lr, lg);
            }
        }
        pp.dump(probe, 0);
    } else {
        // ====================================== REM waves ======================================
        EEG_SETPRIO(EEG_REM_PRIO);      // the younger half of the workgroup
        PhaseProbe<PROBE> pp;
        pp.start();
        int oxw[CT], oh[CT], lt[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ct = own[i] ? wave + 4 * i : 0, nd = remv ? 16 + lg : 16;
            oxw[i] = nd * (3 * H) + ct * 16 + lr;
            oh[i] = nd * H + ct * 16 + lr;
            lt[i] = nd * KAP + ct * 16 + lr;
        }
#pragma unroll
        for (int i = 0; i < CT; ++i)
            if (own[i]) diffuse_rem<M, NR, false>(A, KAP, (wave + 4 * i) * 16, H, Pl, lr, lg);
        for (int t = 0; t < T; ++t) {
            const size_t s = (size_t)t * B + b;
            const float* xw = XW + s * N * (3 * H);
            float xr[CT], xu[CT], xc[CT], ug[CT], g2[2 * CT], c1[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                xr[i] = xw[oxw[i]]; xu[i] = xw[oxw[i] + H]; xc[i] = xw[oxw[i] + 2 * H];
            }
            __syncthreads();                                        // (1)
            pp.mark(0);
            copy_planes(A, Hpl, t);
            f32x4 g4[2 * CT];
            mfma_nodes_rem<2 * CT, KS>(A, KAP, lane, lg, wg, g4);
            pp.mark(1);
            reduce_rem<2 * CT, NR>(g4, lg, g2);
            float* r_t = Rs + s * N * H;
            float* rh_t = RHs + s * N * H;
            float* u_t = Us + s * N * H;
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const float rg = sigmoidf_(g2[i] + xr[i]);
                ug[i] = sigmoidf_(g2[CT + i] + xu[i]);
                if (own[i] && remv) {
                    const float rh = rg * A[lt[i]];
                    A2[lt[i]] = rh;
                    if (save) {
                        r_t[oh[i]] = rg;
                        rh_t[oh[i]] = rh;
                        u_t[oh[i]] = ug[i];
                    }
                }
            }
            __syncthreads();                                        // (1b)
            pp.mark(2);
#pragma unroll
            for (int i = 0; i < CT; ++i)
                if (own[i]) diffuse_rem<M, NR, false>(A2, KAP, (wave + 4 * i) * 16, H, Pl, lr, lg);
            __syncthreads();                                        // (2)
            pp.mark(3);
            copy_planes(A2, RHpl, t);
            f32x4 c4[CT];
            mfma_nodes_rem<CT, KS>(A2, KAP, lane, lg, wc, c4);
            pp.mark(4);
            reduce_rem<CT, NR>(c4, lg, c1);
            float* h_t = Hseq + s * N * H;
            float* c_t = Cs + s * N * H;
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const float pre = c1[i] + xc[i];
                const float c = act == 0 ? tanhf_(pre) : fmaxf(pre, 0.f);
                if (own[i] && remv) {
                    const float h = A[lt[i]], hn = ug[i] * h + (1.f - ug[i]) * c;
                    A[lt[i]] = hn;
                    h_t[oh[i]] = hn;
                    if (save) c_t[oh[i]] = c;
                }
            }
            __syncthreads();                                        // (2b)
            pp.mark(5);
            if (t + 1 < T) {
#pragma unroll
                for (int i = 0; i < CT; ++i)
                    if (own[i]) diffuse_rem<M, NR, false>(A, KAP, (wave + 4 * i) * 16, H, Pl, lr, lg);
            }
        }
        pp.dump(probe, 16);
    }
}

// ================================================================================================
template <int H, int M, int NR, bool PROBE = false>
__global__ __launch_bounds__(512, 2) void seq_bwd_r_kernel(
    const float* __restrict__ Hseq, const float* __restrict__ h0, const float* __restrict__ Rs,
    const float* __restrict__ Us, const float* __restrict__ Cs, const float* __restrict__ dHseq,
    const float* __restrict__ d_at_end, const float* __restrict__ d_at_len, const long long* __restrict__ lengths,
    const float* __restrict__ P, int p_batched, const float* __restrict__ b1p, const float* __restrict__ b2p,
    float* __restrict__ dXW, float* __restrict__ dh0, float* __restrict__ dbias_part, int T, int B, int N, int act,
    long long* probe) {
    using G = SeqGeomR<H, M>;
    constexpr int KAP = G::KAP, KS = G::KS, KGP = G::KGP, KSG = G::KSG, CT = G::CT, NCT = G::NCT;
    EEG_DYN_SMEM(sm);
    float* Pl = sm;
    float* EC = Pl + (M - 1) * kPFloats;    // [20][KAP]  slot 0 = dC, slots m = P_m^T dC
    float* EG = EC + kRRows * KAP;          // [20][KGP]  slot 0 = [dR|dU], slots m = P_m^T [dR|dU]
    float* red = EC;                        // [3H][20] after the loop
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const bool is_tile = wave8 < 4;
    const int wave = wave8 & 3;
    const int b = blockIdx.x;
    const bool remv = lg < NR;

    float w1[CT][KS], w2[CT][KSG];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int ct = wave + 4 * i < NCT ? wave + 4 * i : 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w1[i][ks] = b1p[((size_t)ks * NCT + ct) * 64 + lane];
#pragma unroll
        for (int ks = 0; ks < KSG; ++ks) w2[i][ks] = b2p[((size_t)ks * NCT + ct) * 64 + lane];
    }
    for (int e = tid; e < kRRows * KAP + kRRows * KGP; e += 512) EC[e] = 0.f;
    lds_load_polys(Pl, P, p_batched ? b : 0, M, N);
    const int t_len = (d_at_len != nullptr) ? (lengths != nullptr ? (int)lengths[b] - 1 : T - 1) : -1;
    bool own[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) own[i] = wave + 4 * i < NCT;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const size_t tstride = (size_t)B * N * H;
    const size_t boff = (size_t)b * N * H;
    __syncthreads();

    if (is_tile) {
        // ===================================== TILE waves =====================================
        PhaseProbe<PROBE> pp;
        pp.start();
        float pf0[M - 1][kRNKS];
        load_poly_tile<M, true>(Pl, pf0, lr, lg);
        int oh[CT], oxw[CT], lc[CT], lgt[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ct = own[i] ? wave + 4 * i : 0;
            oh[i] = lr * H + ct * 16 + 4 * lg;
            oxw[i] = lr * (3 * H) + ct * 16 + 4 * lg;
            lc[i] = lr * KAP + ct * 16 + 4 * lg;
            lgt[i] = lr * KGP + ct * 16 + 4 * lg;
        }
        f32x4 dh[CT], sb_r[CT], sb_u[CT], sb_c[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) { dh[i] = zero4; sb_r[i] = zero4; sb_u[i] = zero4; sb_c[i] = zero4; }
        f32x4 nh[CT], nr[CT], nu[CT], nc[CT], ng[CT];
        auto fetch = [&](int t) {
            const size_t so = (size_t)t * tstride + boff;
            const float* hs = t > 0 ? Hseq + (so - tstride) : (h0 != nullptr ? h0 + boff : nullptr);
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int o = oh[i];
                nh[i] = hs != nullptr ? ld4(hs + o) : zero4;
                nr[i] = ld4(Rs + so + o);
                nu[i] = ld4(Us + so + o);
                nc[i] = ld4(Cs + so + o);
                f32x4 g = dHseq != nullptr ? ld4(dHseq + so + o) : zero4;
                if (d_at_end != nullptr && t == T - 1) g += ld4(d_at_end + boff + o);
                if (t == t_len) g += ld4(d_at_len + boff + o);
                ng[i] = g;
            }
        };
        fetch(T - 1);
        for (int t = T - 1; t >= 0; --t) {
            float* dxw = dXW + ((size_t)t * B + b) * N * (3 * H);
            f32x4 hp[CT], rr[CT], dU[CT], dhn[CT], uu[CT], cc[CT], gg[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) { hp[i] = nh[i]; rr[i] = nr[i]; uu[i] = nu[i]; cc[i] = nc[i]; gg[i] = ng[i]; }
            if (t > 0) fetch(t - 1);
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const f32x4 h = hp[i], u = uu[i], c = cc[i];
                const f32x4 g = own[i] ? dh[i] + gg[i] : zero4;
                f32x4 dC, du_;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dc = g[r] * (1.f - u[r]);
                    dC[r] = act == 0 ? dc * (1.f - c[r] * c[r]) : (c[r] > 0.f ? dc : 0.f);
                    du_[r] = g[r] * (h[r] - c[r]) * u[r] * (1.f - u[r]);
                }
                if (own[i]) {
                    st4(EC + lc[i], dC);
                    st4(dxw + oxw[i] + 2 * H, dC);
                    st4(dxw + oxw[i] + H, du_);
                }
                sb_c[i] += dC; sb_u[i] += du_;
                dU[i] = du_; dhn[i] = g * u;
            }
            __syncthreads();                                        // (0b) dC complete (all nodes)
            pp.mark(0);
#pragma unroll
            for (int i = 0; i < CT; ++i)
                if (own[i]) diffuse_tile16<M>(EC, KAP, (wave + 4 * i) * 16, H, pf0, lr, lg);
            __syncthreads();                                        // (1) P_m^T dC complete
            pp.mark(1);
            f32x4 acc[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) acc[i] = zero4;
            mfma_nodes16<CT, KS>(EC, KAP, lr, lg, w1, acc);
            pp.mark(2);
#pragma unroll
            for (int i = 0; i < CT; ++i)
                if (own[i]) {
                    const f32x4 drh = acc[i], rg = rr[i];
                    const f32x4 dR = drh * hp[i] * rg * (1.f - rg);
                    dhn[i] += drh * rg;
                    st4(EG + lgt[i], dR);
                    st4(EG + lgt[i] + H, dU[i]);
                    st4(dxw + oxw[i], dR);
                    sb_r[i] += dR;
                }
            __syncthreads();                                        // (1b) [dR|dU] complete (all nodes)
            pp.mark(3);
#pragma unroll
            for (int i = 0; i < CT; ++i)
                if (own[i]) {
                    diffuse_tile16<M>(EG, KGP, (wave + 4 * i) * 16, 2 * H, pf0, lr, lg);
                    diffuse_tile16<M>(EG, KGP, H + (wave + 4 * i) * 16, 2 * H, pf0, lr, lg);
                }
            __syncthreads();                                        // (2) P_m^T [dR|dU] complete
            pp.mark(4);
            mfma_nodes16<CT, KSG>(EG, KGP, lr, lg, w2, dhn);
            pp.mark(5);
#pragma unroll
            for (int i = 0; i < CT; ++i) dh[i] = dhn[i];
        }
        __syncthreads();                                            // (e1) all waves done with the tiles
        for (int e = tid; e < 3 * H * 20; e += 512) red[e] = 0.f;
        __syncthreads();                                            // (e2)
#pragma unroll
        for (int i = 0; i < CT; ++i)
            if (own[i]) {
                const int col = (wave + 4 * i) * 16;
                if (dh0 != nullptr) st4(dh0 + boff + oh[i], dh[i]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[(0 * H + col + 4 * lg + r) * 20 + lr] = sb_r[i][r];
                    red[(1 * H + col + 4 * lg + r) * 20 + lr] = sb_u[i][r];
                    red[(2 * H + col + 4 * lg + r) * 20 + lr] = sb_c[i][r];
                }
            }
        pp.dump(probe, 8);
    } else {
        // ====================================== REM waves ======================================
        EEG_SETPRIO(EEG_REM_PRIO);
        int oh[CT], oxw[CT], lc[CT], lgt[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ct = own[i] ? wave + 4 * i : 0, nd = remv ? 16 + lg : 16;
            oh[i] = nd * H + ct * 16 + lr;
            oxw[i] = nd * (3 * H) + ct * 16 + lr;
            lc[i] = nd * KAP + ct * 16 + lr;
            lgt[i] = nd * KGP + ct * 16 + lr;
        }
        float dh[CT], sb_r[CT], sb_u[CT], sb_c[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) { dh[i] = 0.f; sb_r[i] = 0.f; sb_u[i] = 0.f; sb_c[i] = 0.f; }
        float nh[CT], nr[CT], nu[CT], nc[CT], ng[CT];
        auto fetch = [&](int t) {
            const size_t so = (size_t)t * tstride + boff;
            const float* hs = t > 0 ? Hseq + (so - tstride) : (h0 != nullptr ? h0 + boff : nullptr);
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int o = oh[i];
                nh[i] = hs != nullptr ? hs[o] : 0.f;
                nr[i] = Rs[so + o];
                nu[i] = Us[so + o];
                nc[i] = Cs[so + o];
                float g = dHseq != nullptr ? dHseq[so + o] : 0.f;
                if (d_at_end != nullptr && t == T - 1) g += d_at_end[boff + o];
                if (t == t_len) g += d_at_len[boff + o];
                ng[i] = g;
            }
        };
        fetch(T - 1);
        for (int t = T - 1; t >= 0; --t) {
            float* dxw = dXW + ((size_t)t * B + b) * N * (3 * H);
            float hp[CT], rr[CT], dU[CT], dhn[CT], uu[CT], cc[CT], gg[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) { hp[i] = nh[i]; rr[i] = nr[i]; uu[i] = nu[i]; cc[i] = nc[i]; gg[i] = ng[i]; }
            if (t > 0) fetch(t - 1);
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const bool ok = own[i] && remv;
                const float g = ok ? dh[i] + gg[i] : 0.f;
                const float dc = g * (1.f - uu[i]);
                const float dC = act == 0 ? dc * (1.f - cc[i] * cc[i]) : (cc[i] > 0.f ? dc : 0.f);
                const float du_ = g * (hp[i] - cc[i]) * uu[i] * (1.f - uu[i]);
                if (ok) {
                    EC[lc[i]] = dC;
                    dxw[oxw[i] + 2 * H] = dC;
                    dxw[oxw[i] + H] = du_;
                }
                sb_c[i] += dC; sb_u[i] += du_;
                dU[i] = du_; dhn[i] = g * uu[i];
            }
            __syncthreads();                                        // (0b)
#pragma unroll
            for (int i = 0; i < CT; ++i)
                if (own[i]) diffuse_rem<M, NR, true>(EC, KAP, (wave + 4 * i) * 16, H, Pl, lr, lg);
            __syncthreads();                                        // (1)
            float drh[CT];
            f32x4 drh4[CT];
            mfma_nodes_rem<CT, KS>(EC, KAP, lane, lg, w1, drh4);
            reduce_rem<CT, NR>(drh4, lg, drh);
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const bool ok = own[i] && remv;
                const float d = ok ? drh[i] : 0.f;
                const float dR = d * hp[i] * rr[i] * (1.f - rr[i]);
                dhn[i] += d * rr[i];
                if (ok) {
                    EG[lgt[i]] = dR;
                    EG[lgt[i] + H] = dU[i];
                    dxw[oxw[i]] = dR;
                }
                sb_r[i] += dR;
            }
            __syncthreads();                                        // (1b)
#pragma unroll
            for (int i = 0; i < CT; ++i)
                if (own[i]) {
                    diffuse_rem<M, NR, true>(EG, KGP, (wave + 4 * i) * 16, 2 * H, Pl, lr, lg);
                    diffuse_rem<M, NR, true>(EG, KGP, H + (wave + 4 * i) * 16, 2 * H, Pl, lr, lg);
                }
            __syncthreads();                                        // (2)
            float d2[CT];
            f32x4 d24[CT];
            mfma_nodes_rem<CT, KSG>(EG, KGP, lane, lg, w2, d24);
            reduce_rem<CT, NR>(d24, lg, d2);
#pragma unroll
            for (int i = 0; i < CT; ++i) dh[i] = (own[i] && remv) ? dhn[i] + d2[i] : 0.f;
        }
        __syncthreads();                                            // (e1)
        for (int e = tid; e < 3 * H * 20; e += 512) red[e] = 0.f;
        __syncthreads();                                            // (e2)
#pragma unroll
        for (int i = 0; i < CT; ++i)
            if (own[i] && remv) {
                const int col = (wave + 4 * i) * 16;
                if (dh0 != nullptr) dh0[boff + oh[i]] = dh[i];
                red[(0 * H + col + lr) * 20 + 16 + lg] = sb_r[i];
                red[(1 * H + col + lr) * 20 + 16 + lg] = sb_u[i];
                red[(2 * H + col + lr) * 20 + 16 + lg] = sb_c[i];
            }
    }
    __syncthreads();                                                // (e3) partial sums staged
    for (int j = tid; j < 3 * H; j += 512) {
        float sacc = 0.f;
#pragma unroll
        for (int q = 0; q < 20; ++q) sacc += red[j * 20 + q];
        dbias_part[(size_t)b * 3 * H + j] = sacc;
    }
}

}  // namespace eeg
