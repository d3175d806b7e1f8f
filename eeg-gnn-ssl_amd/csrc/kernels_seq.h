// Persistent DCGRU sequence kernels (the recurrent half of model/cell.py:182-210 driven by the
// time loop of model/model.py:90-96), one launch per layer and direction.
//
// Design (MI355X-first):
//  * samples are independent, so ONE WORKGROUP OWNS ONE CLIP for the whole sequence: no
//    inter-workgroup traffic, no grid barrier; B workgroups fill the 256 CUs at B = 256.
//  * the recurrent weights (W^h, M*H x 3H fp32 = 147..246 kB) do not fit LDS, so every wave keeps
//    its column slice as MFMA B-fragments IN REGISTERS for all T steps (<= 240 VGPRs of the 512
//    available at one wave per SIMD).
//  * the hidden state / gradient tile, its M hop-diffused copies and the (M-1) hop-polynomial
//    matrices of the clip's graph stay in LDS; the 19-node mix is an fp32 MFMA with the padded
//    32x32 polynomial as A operand.
//  * A fragments come from LDS as ds_read_b128 (four k per lane, K order permuted to match:
//    common.h kperm) and are fetched one quad ahead of the MFMAs that consume them.
//  * the input half of the diffusion convolution (x-part, + biases) is hoisted out of the
//    recurrence (kernels_gemm.h) and arrives as XW (T,B,N,3H) = [r | u | c] pre-activations.
//
// fwd per step:  hops(h) -> G = XW_g + hops(h) Wg^h -> r,u = sigmoid -> hops(r*h)
//                -> C = XW_c + hops(r*h) Wc^h -> c = act(C) -> h' = u*h + (1-u)*c
// bwd per step:  SURVEY.md §9 "Cell backward" with P_m^T adjoint mixes; emits dXW = [dR|dU|dC]
//                per step (consumed afterwards by the hoisted weight-gradient / dX GEMMs).
//
// Element ownership: a lane owns, for each of its column tiles, the C-layout elements
// (row = 16*rt + 4*(lane>>4) + r, col = 16*ct + (lane&15)), rt in {0,1}, r in 0..3.  Rows >= N are
// padding: they are computed (finite garbage stays confined to padding rows) but written to LDS as
// zeros and never stored to HBM.
#pragma once
#include "common.h"
#include "lds_diffuse.h"

namespace eeg {

template <int H, int M>
struct SeqGeom {
    static constexpr int KA = M * H, KAP = lds_stride_q(KA), KS = KA / 4;        // h-wide hop tile
    static constexpr int KG = M * 2 * H, KGP = lds_stride_q(KG), KSG = KG / 4;   // 2H-wide hop tile (bwd)
    static constexpr int NGT = 2 * H / 16, NCT = H / 16;                         // gate / cand col tiles
    static constexpr int GT = ceil_div(NGT, 4), CT = ceil_div(NCT, 4);           // per wave (4 waves)
    static constexpr int US = H + 4;
    static constexpr size_t fwd_lds_floats() { return (size_t)(M - 1) * kPFloats + 2 * 32 * KAP + 32 * US; }
    static constexpr size_t bwd_lds_floats() { return (size_t)(M - 1) * kPFloats + 32 * KAP + 32 * KGP + 3 * H * 4; }
};

// acc[i][rt] += A(32 x 4*NKS, LDS, stride) @ Wfrag[i][.]  for NT column tiles; A fragments are read
// as float4 (k = 16q + 4*(lane>>4) + j) one quad ahead of their use.
template <int NT, int NKS>
__device__ __forceinline__ void mfma_rows32(const float* __restrict__ A, int stride, int lr, int lg,
                                            const float (&w)[NT][NKS], f32x4 (&acc)[NT][2]) {
    static_assert(NKS % 4 == 0, "K must be a multiple of 16");
    const float* p0 = A + lr * stride + 4 * lg;
    const float* p1 = p0 + 16 * stride;
    float4 a0 = *reinterpret_cast<const float4*>(p0);
    float4 a1 = *reinterpret_cast<const float4*>(p1);
#pragma unroll
    for (int q = 0; q < NKS / 4; ++q) {
        float4 n0 = a0, n1 = a1;
        if (q + 1 < NKS / 4) {
            n0 = *reinterpret_cast<const float4*>(p0 + 16 * (q + 1));
            n1 = *reinterpret_cast<const float4*>(p1 + 16 * (q + 1));
        }
        EEG_SCHED_FENCE();      // next quad's fragments are in flight while this quad's MFMAs issue
        const float x0[4] = {a0.x, a0.y, a0.z, a0.w}, x1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                acc[i][0] = mfma16(x0[j], w[i][4 * q + j], acc[i][0]);
                acc[i][1] = mfma16(x1[j], w[i][4 * q + j], acc[i][1]);
            }
        EEG_SCHED_FENCE();
        a0 = n0;
        a1 = n1;
    }
}

// Optional in-kernel phase timer (development aid, enabled through eeg_dcrnn_set_seq_probe):
// lane 0 of every wave accumulates shader-clock cycles per phase and stores them at the end.
struct PhaseProbe {
    long long acc[6];
    long long last;
    bool on;
    __device__ __forceinline__ void start(const long long* p) {
        on = p != nullptr;
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = 0;
        last = on ? cycle_now() : 0;
    }
    __device__ __forceinline__ void mark(int k) {
        if (on) {
            const long long t = cycle_now();
            acc[k] += t - last;
            last = t;
        }
    }
    __device__ __forceinline__ void dump(long long* p, int slot0) {
        if (on && (threadIdx.x & 63) == 0) {
            long long* d = p + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + slot0;
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = acc[i];
        }
    }
};

template <int H, int M>
__global__ __launch_bounds__(256, 1) void seq_fwd_kernel(
    const float* __restrict__ XW, const float* __restrict__ h0, const float* __restrict__ P, int p_batched,
    const float* __restrict__ bhg, const float* __restrict__ bhc,
    float* __restrict__ Hseq, float* __restrict__ Rs, float* __restrict__ Us, float* __restrict__ Cs,
    float* __restrict__ RHs, int T, int B, int N, int act, long long* probe) {
    using G = SeqGeom<H, M>;
    constexpr int KAP = G::KAP, KS = G::KS, GT = G::GT, CT = G::CT, NGT = G::NGT, NCT = G::NCT, US = G::US;
    PhaseProbe pp;
    pp.start(probe);
    EEG_DYN_SMEM(sm);
    float* Pl = sm;
    float* A = Pl + (M - 1) * kPFloats;     // [32][KAP]  slot 0 = h, slots m = P_m h
    float* A2 = A + 32 * KAP;               // [32][KAP]  slot 0 = r*h
    float* Ub = A2 + 32 * KAP;              // [32][US]   update gate
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x;
    const bool save = Rs != nullptr;

    // recurrent weights -> registers (MFMA B fragments), once for all T steps
    float wg[GT][KS], wc[CT][KS];
#pragma unroll
    for (int i = 0; i < GT; ++i) {
        const int ct = wave * GT + i < NGT ? wave * GT + i : 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wg[i][ks] = bhg[((size_t)ks * NGT + ct) * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int ct = wave * CT + i < NCT ? wave * CT + i : 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wc[i][ks] = bhc[((size_t)ks * NCT + ct) * 64 + lane];
    }

    for (int e = tid; e < 2 * 32 * KAP + 32 * US; e += 256) A[e] = 0.f;
    lds_load_polys(Pl, P, p_batched ? b : 0, M, N);
    __syncthreads();
    if (h0 != nullptr)
        for (int e = tid; e < N * H; e += 256) A[(e / H) * KAP + (e % H)] = h0[(size_t)b * N * H + e];
    __syncthreads();

    // per-lane row bookkeeping: rows of the owned elements, clamped copies for safe loads
    int rowv[2][4], rowc[2][4];
    bool valid[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rowv[rt][r] = rt * 16 + 4 * lg + r;
            valid[rt][r] = rowv[rt][r] < N;
            rowc[rt][r] = valid[rt][r] ? rowv[rt][r] : N - 1;
        }

    for (int t = 0; t < T; ++t) {
        const size_t s = (size_t)t * B + b;
        const float* xw = XW + s * N * (3 * H);
        // prefetch this step's hoisted pre-activations (consumed after the diffusion phase);
        // padding rows read a valid row instead of branching
        // (added in the epilogues, so the loads have a whole GEMM to land)
        f32x4 xg[GT][2], xc[CT][2], ag[GT][2], ac[CT][2];
#pragma unroll
        for (int i = 0; i < GT; ++i) {
            const int ct = wave * GT + i < NGT ? wave * GT + i : 0;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                ag[i][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) xg[i][rt][r] = xw[rowc[rt][r] * (3 * H) + ct * 16 + lr];
            }
        }
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ct = wave * CT + i < NCT ? wave * CT + i : 0;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                ac[i][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) xc[i][rt][r] = xw[rowc[rt][r] * (3 * H) + 2 * H + ct * 16 + lr];
            }
        }

        lds_diffuse_tiles<false>(A, KAP, 0, H, H, H, Pl, M, N, 32);
        __syncthreads();                                            // (b) hops(h) complete
        pp.mark(0);

        // gate GEMM: (32 x M*H) @ (M*H x 2H), this wave: GT col tiles x 2 row tiles
        mfma_rows32<GT, KS>(A, KAP, lr, lg, wg, ag);
        pp.mark(1);
#pragma unroll
        for (int i = 0; i < GT; ++i) {
            const int ct = wave * GT + i;
            if (ct < NGT) {                                          // wave-uniform
                const bool is_r = ct < NCT;
                const int col = ct * 16 + lr;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rowv[rt][r];
                        const float g = sigmoidf_(ag[i][rt][r] + xg[i][rt][r]);
                        if (is_r) {
                            const float rh = valid[rt][r] ? g * A[row * KAP + col] : 0.f;
                            A2[row * KAP + col] = rh;
                            if (save && valid[rt][r]) {
                                Rs[(s * N + row) * H + col] = g;
                                RHs[(s * N + row) * H + col] = rh;
                            }
                        } else {
                            Ub[row * US + col - H] = g;
                            if (save && valid[rt][r]) Us[(s * N + row) * H + col - H] = g;
                        }
                    }
            }
        }
        __syncthreads();                                            // (c) r*h and u complete
        pp.mark(2);
        lds_diffuse_tiles<false>(A2, KAP, 0, H, H, H, Pl, M, N, 32);
        __syncthreads();                                            // (d) hops(r*h) complete
        pp.mark(3);

        // candidate GEMM: (32 x M*H) @ (M*H x H), this wave: CT col tiles x 2 row tiles
        mfma_rows32<CT, KS>(A2, KAP, lr, lg, wc, ac);
        pp.mark(4);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ct = wave * CT + i;
            if (ct < NCT) {
                const int col = ct * 16 + lr;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rowv[rt][r];
                        const float pre = ac[i][rt][r] + xc[i][rt][r];
                        const float c = act == 0 ? tanhf_(pre) : fmaxf(pre, 0.f);
                        const float u = Ub[row * US + col], h = A[row * KAP + col];
                        const float hn = valid[rt][r] ? u * h + (1.f - u) * c : 0.f;
                        A[row * KAP + col] = hn;
                        if (valid[rt][r]) {
                            Hseq[(s * N + row) * H + col] = hn;
                            if (save) Cs[(s * N + row) * H + col] = c;
                        }
                    }
            }
        }
        __syncthreads();                                            // (a) h_t complete
        pp.mark(5);
    }
    pp.dump(probe, 0);
}

// lengths: optional int64 (B); d_at_len is added at t = lengths[b]-1, d_at_end at t = T-1.
template <int H, int M>
__global__ __launch_bounds__(256, 1) void seq_bwd_kernel(
    const float* __restrict__ Hseq, const float* __restrict__ h0, const float* __restrict__ Rs,
    const float* __restrict__ Us, const float* __restrict__ Cs, const float* __restrict__ dHseq,
    const float* __restrict__ d_at_end, const float* __restrict__ d_at_len, const long long* __restrict__ lengths,
    const float* __restrict__ P, int p_batched, const float* __restrict__ b1p, const float* __restrict__ b2p,
    float* __restrict__ dXW, float* __restrict__ dh0, float* __restrict__ dbias_part, int T, int B, int N, int act,
    long long* probe) {
    using G = SeqGeom<H, M>;
    constexpr int KAP = G::KAP, KS = G::KS, KGP = G::KGP, KSG = G::KSG, CT = G::CT, NCT = G::NCT;
    PhaseProbe pp;
    pp.start(probe);
    EEG_DYN_SMEM(sm);
    float* Pl = sm;
    float* EC = Pl + (M - 1) * kPFloats;    // [32][KAP]  slot 0 = dC, slots m = P_m^T dC
    float* EG = EC + 32 * KAP;              // [32][KGP]  slot 0 = [dR|dU], slots m = P_m^T [dR|dU]
    float* red = EG + 32 * KGP;             // [3H][4]    bias-gradient reduction scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const int b = blockIdx.x;

    float w1[CT][KS], w2[CT][KSG];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int ct = wave * CT + i < NCT ? wave * CT + i : 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w1[i][ks] = b1p[((size_t)ks * NCT + ct) * 64 + lane];
#pragma unroll
        for (int ks = 0; ks < KSG; ++ks) w2[i][ks] = b2p[((size_t)ks * NCT + ct) * 64 + lane];
    }
    for (int e = tid; e < 32 * KAP + 32 * KGP; e += 256) EC[e] = 0.f;
    lds_load_polys(Pl, P, p_batched ? b : 0, M, N);
    const int t_len = (d_at_len != nullptr) ? (lengths != nullptr ? (int)lengths[b] - 1 : T - 1) : -1;

    int rowv[2][4], rowc[2][4];
    bool valid[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rowv[rt][r] = rt * 16 + 4 * lg + r;
            valid[rt][r] = rowv[rt][r] < N;
            rowc[rt][r] = valid[rt][r] ? rowv[rt][r] : N - 1;
        }

    f32x4 dh[CT][2];
    float sb_r[CT], sb_u[CT], sb_c[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        dh[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dh[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        sb_r[i] = sb_u[i] = sb_c[i] = 0.f;
    }
    __syncthreads();

    const size_t tstride = (size_t)B * N * H;
    // operands of step t are fetched during step t+1 (one step ahead): h_{t-1}, r, u, c and the
    // external gradient of h_t (dHseq + d_at_end + d_at_len); padding rows read a valid row.
    f32x4 nh[CT][2], nr[CT][2], nu[CT][2], nc[CT][2], ng[CT][2];
    auto fetch = [&](int t) {
        const size_t s = (size_t)t * B + b;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ctv = wave * CT + i, ct = ctv < NCT ? ctv : 0, col = ct * 16 + lr;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t e = (s * N + rowc[rt][r]) * H + col, eb = ((size_t)b * N + rowc[rt][r]) * H + col;
                    nh[i][rt][r] = t > 0 ? Hseq[e - tstride] : (h0 != nullptr ? h0[eb] : 0.f);
                    nr[i][rt][r] = Rs[e];
                    nu[i][rt][r] = Us[e];
                    nc[i][rt][r] = Cs[e];
                    float g = dHseq != nullptr ? dHseq[e] : 0.f;
                    if (d_at_end != nullptr && t == T - 1) g += d_at_end[eb];
                    if (t == t_len) g += d_at_len[eb];
                    ng[i][rt][r] = g;
                }
        }
    };
    fetch(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        const size_t s = (size_t)t * B + b;
        f32x4 hp[CT][2], rr[CT][2], dU[CT][2], dhn[CT][2], uu[CT][2], cc[CT][2], gg[CT][2];
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                hp[i][rt] = nh[i][rt]; rr[i][rt] = nr[i][rt]; uu[i][rt] = nu[i][rt]; cc[i][rt] = nc[i][rt]; gg[i][rt] = ng[i][rt];
            }
        if (t > 0) fetch(t - 1);
        // ---- E1: gate blend backward on the owned elements (padding rows zeroed)
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ctv = wave * CT + i, ct = ctv < NCT ? ctv : 0, col = ct * 16 + lr;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = valid[rt][r] && ctv < NCT;
                    const float h = hp[i][rt][r], u = uu[i][rt][r], c = cc[i][rt][r];
                    const float g = ok ? dh[i][rt][r] + gg[i][rt][r] : 0.f;
                    const float dc = g * (1.f - u);
                    const float dC = act == 0 ? dc * (1.f - c * c) : (c > 0.f ? dc : 0.f);
                    const float du_ = g * (h - c) * u * (1.f - u);
                    if (ctv < NCT) EC[rowv[rt][r] * KAP + col] = dC;        // zeros on padding rows
                    if (ok) {
                        dXW[(s * N + rowv[rt][r]) * (3 * H) + 2 * H + col] = dC;
                        dXW[(s * N + rowv[rt][r]) * (3 * H) + H + col] = du_;
                    }
                    sb_c[i] += dC;
                    sb_u[i] += du_;
                    dU[i][rt][r] = du_; dhn[i][rt][r] = g * u;
                }
        }
        __syncthreads();                                            // #1 dC tile complete
        pp.mark(0);
        lds_diffuse_tiles<true>(EC, KAP, 0, H, H, H, Pl, M, N, 32);
        __syncthreads();                                            // #2 P_m^T dC complete
        pp.mark(1);

        // ---- GEMM1: d(r*h) = [P_m^T dC]_m (32 x M*H) @ Wc^h^T (M*H x H)
        f32x4 acc[CT][2];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        mfma_rows32<CT, KS>(EC, KAP, lr, lg, w1, acc);
        pp.mark(2);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int ct = wave * CT + i, col = ct * 16 + lr;
            if (ct < NCT) {                                          // wave-uniform
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rowv[rt][r];
                        const float drh = acc[i][rt][r], rg = rr[i][rt][r];   // exact 0 on padding rows
                        const float dR = drh * hp[i][rt][r] * rg * (1.f - rg);
                        dhn[i][rt][r] += drh * rg;
                        EG[row * KGP + col] = dR;
                        EG[row * KGP + H + col] = dU[i][rt][r];
                        if (valid[rt][r]) dXW[(s * N + row) * (3 * H) + col] = dR;
                        sb_r[i] += dR;
                    }
            }
        }
        __syncthreads();                                            // #3 [dR|dU] tile complete
        pp.mark(3);
        lds_diffuse_tiles<true>(EG, KGP, 0, 2 * H, 2 * H, 2 * H, Pl, M, N, 32);
        __syncthreads();                                            // #4 P_m^T [dR|dU] complete
        pp.mark(4);

        // ---- GEMM2: dh = dhn + [P_m^T dG]_m (32 x M*2H) @ Wg^h^T (M*2H x H)
        mfma_rows32<CT, KSG>(EG, KGP, lr, lg, w2, dhn);
        pp.mark(5);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            dh[i][0] = dhn[i][0];
            dh[i][1] = dhn[i][1];
        }
    }

    // ---- epilogue: dh0 and the per-clip bias-gradient partial sums
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int ct = wave * CT + i, col = ct * 16 + lr;
        if (ct < NCT) {
            if (dh0 != nullptr) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (valid[rt][r]) dh0[((size_t)b * N + rowv[rt][r]) * H + col] = dh[i][rt][r];
            }
            red[(0 * H + col) * 4 + lg] = sb_r[i];
            red[(1 * H + col) * 4 + lg] = sb_u[i];
            red[(2 * H + col) * 4 + lg] = sb_c[i];
        }
    }
    __syncthreads();
    for (int j = tid; j < 3 * H; j += 256)
        dbias_part[(size_t)b * 3 * H + j] = (red[j * 4] + red[j * 4 + 1]) + (red[j * 4 + 2] + red[j * 4 + 3]);
    pp.dump(probe, 8);
}

}  // namespace eeg
