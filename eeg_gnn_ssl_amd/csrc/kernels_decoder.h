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
#include <type_traits>

#include "common.h"
#include "kernels_seq.h"
#include "lds_diffuse.h"
#include "nnq_order.h"

namespace eeg {

struct DecLayerPtrs {
    const float *bxq, *bias, *bhg, *bhc;                      // weight packs of the layer's cell (kernels_pack.h; bxq: the x-part in quad order)
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
    const int* teacher_dev;            // nullable DEVICE int32[T]: when set, the flags are read from it at launch instead (a captured
                                       // graph then replays with whatever eeg_dcrnn_teacher_flags drew in front of it)
    int p_batched, T, B, N, Dout, L, act;
    // nn.Dropout in front of the projection (model.py:191, training): the projection reads drop(h_top_t) = mask * h_top_t; the
    // recurrence keeps h_top_t.  Element ((t*B + b)*N + n)*H + h of the (T,B,N,H) top outputs takes word e % 4 of Philox counter
    // rng_used[1] + e / 4 under key rng_used[0] (common.h).  hd (T,B,N,H): the dropped rows, the A operand of dW_p in the backward.
    DropCfg drop;
    const unsigned long long* rng_used;
    float* hd;
    long long* probe;             // phase cycles (development builds with -DEEG_DEC_PROBE: tools/dec_probe.py), else unused
};
#if defined(EEG_DEV) && defined(EEG_DEC_PROBE)
constexpr bool kDecProbe = true;
#else
constexpr bool kDecProbe = false;
#endif
// shader-clock cycles per phase, 16 slots per wave (layer 0: slots 0..7, layers above: 8..15), as PhaseProbe (kernels_seq.h)
template <bool ON>
struct DecProbe {
    long long acc[16];
    long long last;
    __device__ __forceinline__ void start() {
        if (ON) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0;
            last = cycle_now();
        }
    }
    __device__ __forceinline__ void mark(int k, int l = 0) {
        if (ON) {
            const long long t = cycle_now();
            if (l == 0) acc[k] += t - last; else acc[8 + k] += t - last;
            last = t;
        }
    }
    __device__ __forceinline__ void dump(long long* p) {
        if (ON && p != nullptr && (threadIdx.x & 63) == 0) {
            long long* d = p + ((size_t)blockIdx.x * 4 + ((threadIdx.x >> 6) & 3)) * 32;
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = acc[i];
        }
    }
};

// ---- GEMMs with streamed weights -------------------------------------------------------------------------------
// acc[i][0] += (weight tile wt[i])^T x X^T for nodes 0..15, issued transposed like mfma_nodes32 (lane: node lr, 4 consecutive
// columns); remainder nodes 16..19 on v_mfma_f32_4x4x1, reduced in registers to ONE element per lane (mfma_nodes32 L1): lane
// (lr, lg) gets acc[i][1][0] += out[node 16 + lg][column lr of tile wt[i]]; components 1..3 of acc[i][1] are not touched.
// Plain K order (pack index ks = k/4, k = slot*KPP*4 + f): the operand tile has `nslots` hop slots of `slotw` columns of
// which the first 4*KPP are real; SWZ: the tile is an XOR-swizzled state tile (lds_sw), else a plain [rows][stride] one.
// The weights of group g+1 (D k-steps x NT tiles, one coalesced dword per lane each) are requested before the MFMAs
// of group g.  KPP % D == 0.
// The first weight group can be requested by the caller long before the call (plain_wload into `wa`, then PRE = true):
// between two GEMMs of a step there is elementwise work, a node mix and a barrier for the L2 latency to hide behind.
// The last hop slot may come from a second tile (tile_last != nullptr: columns 0.. of that tile).
template <int NT, int D>
__device__ __forceinline__ void plain_wload(const float* __restrict__ wp, int nct_total, const int (&wt)[NT], int lane, int g,
                                            float (&w)[D][NT]) {
    const wbuf_t wb = make_wbuf(wp);
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int i = 0; i < NT; ++i) w[d][i] = wbuf_ld(wb, wt[i] * 64 + lane, (g * D + d) * nct_total * 64);
}
template <int NT, int D, bool SWZ, bool PRE = false, bool XP = true>
__device__ __forceinline__ void gemm_stream_plain(const float* __restrict__ tile, int stride, int slotw, int kpp, int nslots,
                                                  const float* __restrict__ wp, int nct_total, const int (&wt)[NT],
                                                  int lane, int lr, int lg, f32x4 (&acc)[NT][2],
                                                  float (&wa)[D][NT], const float* __restrict__ tile_last = nullptr,
                                                  int stride_last = 0) {
    const int ngroups = nslots * kpp / D;
    const int row1 = 16 + (lane & 3);
    f32x4 rem[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rem[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int slot = 0, f0 = 0;
    auto xload = [&](float (&x0)[D], float (&x1)[D]) {            // fragments of the next group; advances (slot, f0)
        const bool last = tile_last != nullptr && slot == nslots - 1;
        const float* tp = last ? tile_last : tile;
        const int st = last ? stride_last : stride, cb = last ? 0 : slot * slotw;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int col = cb + 4 * (f0 + d) + lg;
            x0[d] = tp[SWZ ? lds_sw(lr, col, st) : lr * st + col];
            x1[d] = tp[SWZ ? lds_sw(row1, col, st) : row1 * st + col];
        }
        f0 += D;
        if (f0 == kpp) { f0 = 0; ++slot; }
    };
    // One group = all its 16x16x4 MFMAs, then all its 4x4x1 MFMAs: a 4x4x1 behind a 16x16x4 costs ~19 cycles instead of 8, up
    // to ~43 cycles per run however long (tools/micro/chain_lab.hip) -- alternating the shapes k-step by k-step paid that on
    // every one.  A lone column tile alternates between two 16x16x4 chains (a dependent chain issues every 52 cycles, not 32).
    f32x4 alt = {0.f, 0.f, 0.f, 0.f};
    auto mac = [&](const float (&w)[D][NT], const float (&x0)[D], const float (&x1)[D]) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (NT == 1 && (d & 1)) alt = mfma16(w[d][i], x0[d], alt);
                else acc[i][0] = mfma16(w[d][i], x0[d], acc[i][0]);
            }
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int i = 0; i < NT; ++i) rem[i][d & 3] = mfma4(x1[d], w[d][i], rem[i][d & 3]);
    };
    float wb[D][NT], xa0[D], xa1[D];
    if (!PRE) plain_wload<NT, D>(wp, nct_total, wt, lane, 0, wa);
    if constexpr (XP) {                                            // operand fragments one group ahead of their MFMAs
        float xb0[D], xb1[D];
        xload(xa0, xa1);
        for (int g = 0; g < ngroups; g += 2) {
            if (g + 1 < ngroups) {
                plain_wload<NT, D>(wp, nct_total, wt, lane, g + 1, wb);
                xload(xb0, xb1);
            }
            EEG_SCHED_FENCE();
            mac(wa, xa0, xa1);
            if (g + 1 < ngroups) {
                if (g + 2 < ngroups) {
                    plain_wload<NT, D>(wp, nct_total, wt, lane, g + 2, wa);
                    xload(xa0, xa1);
                }
                EEG_SCHED_FENCE();
                mac(wb, xb0, xb1);
            }
        }
    } else {
        for (int g = 0; g < ngroups; g += 2) {
            if (g + 1 < ngroups) plain_wload<NT, D>(wp, nct_total, wt, lane, g + 1, wb);
            xload(xa0, xa1);
            EEG_SCHED_FENCE();
            mac(wa, xa0, xa1);
            if (g + 1 < ngroups) {
                if (g + 2 < ngroups) plain_wload<NT, D>(wp, nct_total, wt, lane, g + 2, wa);
                xload(xa0, xa1);
                EEG_SCHED_FENCE();
                mac(wb, xa0, xa1);
            }
        }
    }
    if (NT == 1) acc[0][0] += alt;
    // remainder: reduce-scatter in registers, one element per lane (mfma_nodes32 L1; common.h rem4_reduce)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        f32x4 t = (rem[i][0] + rem[i][1]) + (rem[i][2] + rem[i][3]);
        EEG_PIN(t);
        acc[i][1][0] += rem4_reduce(t);
    }
}
template <int NT, int D, bool SWZ>
__device__ __forceinline__ void gemm_stream_plain(const float* __restrict__ tile, int stride, int slotw, int kpp, int nslots,
                                                  const float* __restrict__ wp, int nct_total, const int (&wt)[NT],
                                                  int lane, int lr, int lg, f32x4 (&acc)[NT][2]) {
    float wa[D][NT];
    gemm_stream_plain<NT, D, SWZ, false, false>(tile, stride, slotw, kpp, nslots, wp, nct_total, wt, lane, lr, lg, acc, wa);
}

// Same for the recurrent packs (quad-permuted K order, ds_read_b128 fragments of a swizzled state tile): NQ quads
// (compile time), weights requested PD quads ahead; the operand fragments one quad ahead.  PRE: the caller has already
// requested the first PD quads (quad_prefetch into `w`).
template <int NT, int NQ, int PD, int NTW = NT>
__device__ __forceinline__ void quad_prefetch(const float* __restrict__ wp, int nct_total, const int (&wt)[NT], int lane,
                                              float (&w)[PD + 1][4][NTW]) {
    const wbuf_t wb = make_wbuf(wp);
#pragma unroll
    for (int q = 0; q < PD && q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) w[q % (PD + 1)][j][i] = wbuf_ld(wb, wt[i] * 64 + lane, (4 * q + j) * nct_total * 64);
}
template <int NT, int NQ, int PD, bool PRE = false, int NTW = NT>
__device__ __forceinline__ void gemm_stream_quad(const float* __restrict__ tile, int stride, const float* __restrict__ wp,
                                                 int nct_total, const int (&wt)[NT], int lane, int lr, int lg,
                                                 f32x4 (&acc)[NT][2], float (&w)[PD + 1][4][NTW]) {
    const int s0 = lg ^ sigma4(lr), s1 = lg ^ sigma4(lane & 3);
    const float* p0 = tile + lr * stride;
    const float* p1 = tile + (16 + (lane & 3)) * stride;
    auto frag = [&](const float* rowp, int sx, int q) {
        return *reinterpret_cast<const float4*>(rowp + 64 * (q >> 2) + 4 * ((4 * (q & 3)) ^ sx));
    };
    const wbuf_t wb = make_wbuf(wp);
    auto wload = [&](int q, float (&dst)[4][NTW]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) dst[j][i] = wbuf_ld(wb, wt[i] * 64 + lane, (4 * q + j) * nct_total * 64);
    };
    f32x4 rem[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rem[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!PRE) quad_prefetch<NT, NQ, PD, NTW>(wp, nct_total, wt, lane, w);
    f32x4 alt = {0.f, 0.f, 0.f, 0.f};            // second 16x16x4 chain of a lone column tile
    float4 a0 = frag(p0, s0, 0), a1 = frag(p1, s1, 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q + PD < NQ) wload(q + PD, w[(q + PD) % (PD + 1)]);
        float4 n0 = a0, n1 = a1;
        if (q + 1 < NQ) {
            n0 = frag(p0, s0, q + 1);
            n1 = frag(p1, s1, q + 1);
        }
        const float x0[4] = {a0.x, a0.y, a0.z, a0.w}, x1[4] = {a1.x, a1.y, a1.z, a1.w};
        EEG_SCHED_FENCE();
#pragma unroll
        for (int j = 0; j < 4; ++j)              // the quad's 16x16x4 MFMAs, then its 4x4x1 MFMAs (see gemm_stream_plain)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (NT == 1 && (j & 1)) alt = mfma16(w[q % (PD + 1)][j][i], x0[j], alt);
                else acc[i][0] = mfma16(w[q % (PD + 1)][j][i], x0[j], acc[i][0]);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) rem[i][j] = mfma4(x1[j], w[q % (PD + 1)][j][i], rem[i][j]);
        a0 = n0;
        a1 = n1;
    }
    if (NT == 1) acc[0][0] += alt;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        f32x4 t = (rem[i][0] + rem[i][1]) + (rem[i][2] + rem[i][3]);
        EEG_PIN(t);
        acc[i][1][0] += rem4_reduce(t);
    }
}
template <int NT, int NQ, int PD>
__device__ __forceinline__ void gemm_stream_quad(const float* __restrict__ tile, int stride, const float* __restrict__ wp,
                                                 int nct_total, const int (&wt)[NT], int lane, int lr, int lg,
                                                 f32x4 (&acc)[NT][2]) {
    float w[PD + 1][4][NT];
    gemm_stream_quad<NT, NQ, PD, false>(tile, stride, wp, nct_total, wt, lane, lr, lg, acc, w);
}

// The x-part of a cell with the weights in the quad pack of gemm_nnr_kernel (kernels_pack.h bxq: chunk order of make_nnq_order over
// nseg hop slots x F features; [(c * nct_total + ct) * 64 + lane][4], zero chunks appended up to a multiple of 4): per 16-deep
// chunk ONE 16-byte weight load per lane and column tile and ONE ds_read_b128 per node tile feed four k-steps -- a quarter of the
// vector-memory and LDS instructions of gemm_stream_plain, whose 4-byte loads (3 + 2 per k-step beside 6 MFMAs, ~200 registers of
// look-ahead at 25 k-steps per group) left the x-part at half the MFMA rate (profiles/r04_i_dec_probe.txt).
// The stream is BRANCH-FREE: groups of 4 chunks (= the weight ring: 3 chunks requested ahead), every body issues its loads
// unconditionally (past the end: the last chunk again, into a slot nothing reads any more), so the compiler's own s_waitcnt
// counts are the steady-state ones; with a conditional load in the body it waited for ALL loads, i.e. one L2 round trip per chunk.
// Operand tile, SWZ = true: the XOR-swizzled state tile of the layer below (slots of 64 = 4 whole chunks, no tail: chunk c is
// quad c of gemm_stream_quad).  SWZ = false: the plain step-input tile (slots of slotw columns, F real); the column of lane
// group lg's 16-byte piece of chunk c comes from a table in LDS (nnq_col_table: tail chunks gather the leftover pieces of the
// slots; padding chunks and missing pieces point at column 0: zero weights, any finite operand does).
constexpr int kNnqPD = 3;
__host__ __device__ inline int nnq_padded_chunks(int nseg, int F) { return round_up(make_nnq_order(nseg, F).nch, 4); }
// tab[c * 4 + g], c < nnq_padded_chunks: float column of piece g of chunk c in a tile with hop slots of slotw columns
__device__ __forceinline__ void nnq_col_table(int* tab, int nseg, int F, int slotw) {
    const NnqOrder ko = make_nnq_order(nseg, F);
    const int n = 4 * round_up(ko.nch, 4);
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int c = e >> 2, g = e & 3;
        int col = 0;
        if (c < ko.nmain) {
            const int seg = c / ko.a;
            col = seg * slotw + (c - seg * ko.a) * 16 + 4 * g;
        } else {
            const int tp = (c - ko.nmain) * 4 + g;
            if (tp < ko.nseg * ko.b) col = (tp / ko.b) * slotw + ko.a * 16 + (tp % ko.b) * 4;
        }
        tab[e] = col;
    }
}
template <int NT>
__device__ __forceinline__ void nnq_prefetch(const float* __restrict__ Bq, int nct_total, const int (&wt)[NT], int lane,
                                             f32x4 (&w)[kNnqPD + 1][NT]) {
    const wbuf_t wb = make_wbuf(Bq);
#pragma unroll
    for (int c = 0; c < kNnqPD; ++c)             // (a pack has at least 4 chunks)
#pragma unroll
        for (int i = 0; i < NT; ++i) w[c][i] = wbuf_ld4(wb, (unsigned)(wt[i] * 64 + lane) * 4u, (unsigned)(c * nct_total) * 256u);
}
// nchp = nnq_padded_chunks; the first 3 chunks of weights are in w[0..2] (nnq_prefetch); tab: nnq_col_table (SWZ = false only)
template <int NT, bool SWZ>
__device__ __forceinline__ void gemm_stream_nnq(const float* __restrict__ tile, int stride, const int* __restrict__ tab, int nchp,
                                                const float* __restrict__ Bq, int nct_total, const int (&wt)[NT],
                                                int lane, int lr, int lg, f32x4 (&acc)[NT][2], f32x4 (&w)[kNnqPD + 1][NT]) {
    constexpr int PD = kNnqPD, R = PD + 1;
    const int row1 = 16 + (lane & 3);
    const wbuf_t wb = make_wbuf(Bq);
    unsigned wv[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) wv[i] = (unsigned)(wt[i] * 64 + lane) * 4u;
    // SWZ: piece 4j + lg of quad-group Q of row r sits at 64 Q + 4 ((4j + lg) ^ sigma4(r)); the 4 j-offsets are lane constants
    const float* p0 = tile + lr * stride;
    const float* p1 = tile + row1 * stride;
    int k0[4], k1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        k0[j] = SWZ ? 4 * ((4 * j + lg) ^ sigma4(lr)) : 0;
        k1[j] = SWZ ? 4 * ((4 * j + lg) ^ sigma4(lane & 3)) : 0;
    }
    f32x4 rem[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rem[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 alt = {0.f, 0.f, 0.f, 0.f};            // second 16x16x4 chain of a lone column tile
    int tc = SWZ ? 0 : tab[lg], tn = SWZ ? 0 : tab[4 + lg];         // piece columns of chunks 0 and 1 (plain tile)
    float4 a0 = *reinterpret_cast<const float4*>(p0 + (SWZ ? k0[0] : tc));
    float4 a1 = *reinterpret_cast<const float4*>(p1 + (SWZ ? k1[0] : tc));
    const int ngroups = nchp / R;
    for (int g = 0; g < ngroups; ++g) {
        static_for<0, R>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value, slot = j, slot_n = (j + PD) % R;
            const int c = g * R + j;
            const int cl = c + PD < nchp ? c + PD : nchp - 1;        // (scalar)
#pragma unroll
            for (int i = 0; i < NT; ++i) w[slot_n][i] = wbuf_ld4(wb, wv[i], (unsigned)(cl * nct_total) * 256u);
            // fragments of chunk c + 1 (past the end: chunk nchp - 1 again)
            float4 n0, n1;
            if constexpr (SWZ) {
                const int cn = c + 1 < nchp ? c + 1 : nchp - 1, qn = 64 * (cn >> 2);
                constexpr int jn = (j + 1) % R;
                n0 = *reinterpret_cast<const float4*>(p0 + qn + k0[jn]);
                n1 = *reinterpret_cast<const float4*>(p1 + qn + k1[jn]);
            } else {
                n0 = *reinterpret_cast<const float4*>(p0 + tn);
                n1 = *reinterpret_cast<const float4*>(p1 + tn);
                const int c2 = c + 2 < nchp ? c + 2 : nchp - 1;
                tn = tab[c2 * 4 + lg];
            }
            const float x0[4] = {a0.x, a0.y, a0.z, a0.w}, x1[4] = {a1.x, a1.y, a1.z, a1.w};
            EEG_SCHED_FENCE();
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    if (NT == 1 && (jj & 1)) alt = mfma16(w[slot][i][jj], x0[jj], alt);
                    else acc[i][0] = mfma16(w[slot][i][jj], x0[jj], acc[i][0]);
                }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < NT; ++i) rem[i][jj] = mfma4(x1[jj], w[slot][i][jj], rem[i][jj]);
            a0 = n0;
            a1 = n1;
        });
    }
    if (NT == 1) acc[0][0] += alt;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        f32x4 t = (rem[i][0] + rem[i][1]) + (rem[i][2] + rem[i][3]);
        EEG_PIN(t);
        acc[i][1][0] += rem4_reduce(t);
    }
}

constexpr int kDecRows = 20;     // node rows of the LDS tiles (montages of at most 20 nodes)

// LDS floats of dec_fwd_persist_kernel<64, M>
__host__ __device__ constexpr size_t dec_fwd_lds_floats(int M, int L, int Dout) {
    const int H = 64, KAP = M * H, XS = lds_stride_q(M * round_up(Dout, 16));
    return (size_t)(M - 1) * kPFloats + (size_t)L * kDecRows * KAP + (size_t)kDecRows * (XS > KAP ? XS : KAP)
           + (size_t)kDecRows * 64       // + the dropped copy of the top state the projection reads (swizzled [rows][64])
           + 4 * (size_t)nnq_padded_chunks(M, Dout);   // + the piece-column table of the layer-0 x-part (gemm_stream_nnq)
}

template <int H, int M>
__global__ __launch_bounds__(256, 1) void dec_fwd_persist_kernel(DecFwdArgs a) {
    unsigned long long teacher_mask = a.teacher_mask;
    if (a.teacher_dev != nullptr) {                 // flags drawn on the device (uniform scalar loads, once per launch)
        teacher_mask = 0;
        for (int t = 0; t < a.T; ++t)
            if (a.teacher_dev[t] != 0) teacher_mask |= 1ull << t;
    }
    static_assert(H == 64, "one column tile per wave");
    constexpr int NKS = 5, ROWS = kDecRows, KAP = M * H, NCT = H / 16, NGT = 2 * NCT, NQ = M * H / 16;
    EEG_DYN_SMEM(sm);
    const int T = a.T, B = a.B, N = a.N, Dout = a.Dout, L = a.L, act = a.act;
    // X0 row stride = 4 mod 64 dwords: the ds_read_b32 fragment reads of the layer-0 input GEMM (lane (lr, lg): row lr,
    // column c + lg) then cover the 64 banks exactly (M*FP = 560 gave 4-way conflicts: 7.8 % of the wave cycles)
    const int FP = round_up(Dout, 16), XS = lds_stride_q(M * FP), XK = XS > KAP ? XS : KAP;
    float* Pl = sm;
    float* A0 = Pl + (M - 1) * kPFloats;            // L state tiles [ROWS][KAP]: slot 0 = h^l, slots m = P_m h^l
    float* XA = A0 + L * ROWS * KAP;                // step-input tile X0 [ROWS][XS] (plain)  |  r*h tile A2 [ROWS][KAP] (swizzled)
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), lr = lane & 15, lg = lane >> 4;
    float* HD = XA + ROWS * XK;                     // [ROWS][64] swizzled: drop(h^{L-1}_t), written and read only when a.drop.on
    const int ct = wave, col = ct * 16 + 4 * lg;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int nct_o = ceil_div(Dout, 16);
    const size_t xstep = (size_t)B * N * Dout;
    const DropCfg drop = a.drop;
    const unsigned long long dseed = drop.on ? a.rng_used[0] : 0ull, doff = drop.on ? a.rng_used[1] : 0ull;
    const int wt3[3] = {ct, NCT + ct, 2 * NCT + ct};
    f32x4 wx[kNnqPD + 1][3];       // weight ring of the x-part (gemm_stream_nnq)
    int* XT = reinterpret_cast<int*>(HD + ROWS * 64);               // piece columns of the layer-0 x-part chunks in X0
    const int nchp0 = nnq_padded_chunks(M, Dout), nchp1 = nnq_padded_chunks(M, H);
    nnq_col_table(XT, M, Dout, FP);                                 // (first read behind the barriers of the clip set-up)
    float wpp[8][2];               // first weight group (8 of the 16 k-steps) of the projection; 16: spills at M = 5
    bool xpre = false;
    DecProbe<kDecProbe> pp;        // 0 input node mix + planes, 1 x-part GEMM, 2 barrier, 3 gate GEMM, 4 gate epilogue + node mix + barrier,
    pp.start();                    // 5 candidate GEMM, 6 its epilogue + node mix + barrier, 7 projection + barrier

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < L * ROWS * KAP + ROWS * XK; e += 256) A0[e] = 0.f;
        lds_load_polys(Pl, a.P, a.p_batched ? b : 0, M, N);
        __syncthreads();
        float pf[poly_slots<M, NKS>()][NKS];
        load_poly_frags<M, NKS, false>(Pl, pf, lr, lg);
        const bool valid[2] = {lr < N, 16 + lr < N};
        const int nodec[2] = {valid[0] ? lr : N - 1, valid[1] ? 16 + lr : N - 1};
        const int oh[2] = {nodec[0] * H + col, nodec[1] * H + col};
        // remainder nodes 16..19: one element per lane -- lane (lr, lg) <-> node 16 + lg, column lr of the wave's column tile
        const int node1 = 16 + lg, col1 = ct * 16 + lr;
        const bool valid1 = node1 < N;
        const int oh1 = (valid1 ? node1 : N - 1) * H + col1;
        const int l1 = lds_sw(node1, col1, KAP);
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
        pp.mark(7);
        for (int t = 0; t < T; ++t) {
            const size_t s = (size_t)t * B + b;
            // ---- hop rows of the step input (X0 slot 0 holds it: zeros at t = 0, GO symbol) -> slots 1..M-1, and to planes0
            // (register-resident polynomial fragments, one 16-column tile at a time as for the state tiles; the generic work-list
            //  routine lds_diffuse_tiles + a copy loop took 15.5 k of the 115 k cycles of a step: profiles/r04_i_dec_probe.txt)
            nnq_prefetch<3>(a.l[0].bxq, 3 * NCT, wt3, lane, wx);
            xpre = true;
            for (int j = wave; j < FP / 16; j += 4)
                lds_diffuse_tile_plain<M, NKS, ROWS>(XA, XS, j * 16, FP, pf, lr, lg, a.planes0 + s * N * Dout, a.planes0_stride, Dout, N);
            __syncthreads();
            pp.mark(0);
            for (int l = 0; l < L; ++l) {
                const DecLayerPtrs& lp = a.l[l];
                float* Al = A0 + l * ROWS * KAP;
                float* A2 = XA;
                f32x4 ag[3][2];          // pre-activations of this wave's r, u, c tiles: x-part + bias, then + h-part
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    ag[i][0] = ld4(lp.bias + wt3[i] * 16 + 4 * lg);
                    ag[i][1] = (f32x4){lp.bias[wt3[i] * 16 + lr], 0.f, 0.f, 0.f};
                }
                {
                    if (!xpre) nnq_prefetch<3>(lp.bxq, 3 * NCT, wt3, lane, wx);
                    xpre = false;
                    if (l == 0) gemm_stream_nnq<3, false>(XA, XS, XT, nchp0, lp.bxq, 3 * NCT, wt3, lane, lr, lg, ag, wx);
                    else gemm_stream_nnq<3, true>(A0 + (l - 1) * ROWS * KAP, KAP, nullptr, nchp1, lp.bxq, 3 * NCT, wt3, lane, lr, lg, ag, wx);
                }
                // the first quads of the gate / candidate weights are requested before the barrier / the epilogue in front of their GEMM
                constexpr int PDG = NQ < 6 ? NQ : 6, PDC = NQ < 10 ? NQ : 10;
                const int wt2[2] = {ct, NCT + ct}, wt1[1] = {ct};
                float wqg[PDG + 1][4][2], wqc[PDC + 1][4][1];
                quad_prefetch<2, NQ, PDG>(lp.bhg, NGT, wt2, lane, wqg);
                pp.mark(1, l);
                __syncthreads();                                     // layer 0: every wave has read X0 (A2 aliases it)
                pp.mark(2, l);
                // gate h-part: hops(h^l) x Wg^h
                {
                    f32x4 g2[2][2] = {{ag[0][0], ag[0][1]}, {ag[1][0], ag[1][1]}};
                    gemm_stream_quad<2, NQ, PDG, true>(Al, KAP, lp.bhg, NGT, wt2, lane, lr, lg, g2, wqg);
                    ag[0][0] = g2[0][0]; ag[0][1] = g2[0][1]; ag[1][0] = g2[1][0]; ag[1][1] = g2[1][1];
                }
                quad_prefetch<1, NQ, PDC>(lp.bhc, NCT, wt1, lane, wqc);
                pp.mark(3, l);
                f32x4 ug0;
                float ug1;
                {
                    const f32x4 rg = sigmoid4_(ag[0][0]), u = sigmoid4_(ag[1][0]);
                    ug0 = u;
                    f32x4 rh = rg * ld4(Al + lds_sw(lr, col, KAP));
                    rh = valid[0] ? rh : zero4;
                    st4(A2 + lds_sw(lr, col, KAP), rh);
                    if (valid[0]) {
                        st4(lp.rs + s * N * H + oh[0], rg);
                        st4(lp.rhs + s * N * H + oh[0], rh);
                        st4(lp.us + s * N * H + oh[0], u);
                    }
                    const float rg1 = sigmoidf_(ag[0][1][0]), u1 = sigmoidf_(ag[1][1][0]);
                    ug1 = u1;
                    const float rh1 = valid1 ? rg1 * Al[l1] : 0.f;
                    A2[l1] = rh1;
                    if (valid1) {
                        lp.rs[s * N * H + oh1] = rg1;
                        lp.rhs[s * N * H + oh1] = rh1;
                        lp.us[s * N * H + oh1] = u1;
                    }
                }
                EEG_WAVE_SYNC();
                lds_diffuse_tile<M, NKS, ROWS>(A2, KAP, ct * 16, H, pf, lr, lg, lp.rpl + s * N * H, a.hplane_stride, N);
                __syncthreads();                                     // hops(r*h) complete
                pp.mark(4, l);
                // candidate h-part: hops(r*h) x Wc^h
                {
                    f32x4 c1[1][2] = {{ag[2][0], ag[2][1]}};
                    gemm_stream_quad<1, NQ, PDC, true>(A2, KAP, lp.bhc, NCT, wt1, lane, lr, lg, c1, wqc);
                    ag[2][0] = c1[0][0]; ag[2][1] = c1[0][1];
                }
                pp.mark(5, l);
                {
                    const f32x4 u = ug0, h = ld4(Al + lds_sw(lr, col, KAP));
                    const f32x4 c = act == 0 ? tanh4_(ag[2][0]) : relu4_(ag[2][0]);
                    f32x4 hn = u * h + (1.f - u) * c;
                    hn = valid[0] ? hn : zero4;
                    st4(Al + lds_sw(lr, col, KAP), hn);
                    if (valid[0]) {
                        st4(lp.hext + (s + B) * N * H + oh[0], hn);              // hext slot t+1
                        st4(lp.cs + s * N * H + oh[0], c);
                    }
                    const float h1 = Al[l1], pre1 = ag[2][1][0];
                    const float c1 = act == 0 ? tanhf_(pre1) : fmaxf(pre1, 0.f);
                    const float hn1 = valid1 ? ug1 * h1 + (1.f - ug1) * c1 : 0.f;
                    Al[l1] = hn1;
                    if (valid1) {
                        lp.hext[(s + B) * N * H + oh1] = hn1;
                        lp.cs[s * N * H + oh1] = c1;
                    }
                    if (drop.on && l == L - 1) {                                 // what the projection reads (model.py:191)
                        const f32x4 hd = hn * dropout_mask4(dseed, doff, (s * N * H + oh[0]) >> 2, drop.thr, drop.scale);
                        st4(HD + lds_sw(lr, col, 64), hd);
                        if (valid[0]) st4(a.hd + s * N * H + oh[0], hd);
                        const float hd1 = hn1 * dropout_mask1(dseed, doff, s * N * H + oh1, drop.thr, drop.scale);
                        HD[lds_sw(node1, col1, 64)] = hd1;
                        if (valid1) a.hd[s * N * H + oh1] = hd1;
                    }
                }
                EEG_WAVE_SYNC();
                if (l + 1 < L) {                                     // the next layer's first x-part weights, behind this node mix and barrier
                    nnq_prefetch<3>(a.l[l + 1].bxq, 3 * NCT, wt3, lane, wx);
                    xpre = true;
                } else if (wave < nct_o) {                           // the projection's weights (this wave's tiles wave, wave + 4)
                    const int wtp[2] = {wave, wave + 4 < nct_o ? wave + 4 : wave};
                    plain_wload<2, 8>(a.ppack, nct_o, wtp, lane, 0, wpp);
                }
                lds_diffuse_tile<M, NKS, ROWS>(Al, KAP, ct * 16, H, pf, lr, lg, lp.hpl + (s + B) * N * H, a.hplane_stride, N);
                __syncthreads();                                     // h^l of this step and its hop rows complete
                pp.mark(6, l);
            }
            // ---- projection (model.py:188-190) and the next step's input (model.py:194-200)
            {
                float* Atop = A0 + (L - 1) * ROWS * KAP;
                const bool tf = ((teacher_mask >> t) & 1ull) != 0;
                for (int j0 = wave; j0 < nct_o; j0 += 8) {              // this wave's tiles j0 and j0 + 4
                    const bool two = j0 + 4 < nct_o;
                    f32x4 po[2][2];
                    const int wt2[2] = {j0, two ? j0 + 4 : j0};
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        po[i][0] = ld4(a.pbias + wt2[i] * 16 + 4 * lg);
                        po[i][1] = (f32x4){a.pbias[wt2[i] * 16 + lr], 0.f, 0.f, 0.f};
                    }
                        if (j0 == wave)         // first pair of tiles: weights requested in front of the last barrier
                        gemm_stream_plain<2, 8, true, true, false>(drop.on ? HD : Atop, drop.on ? 64 : KAP, H, H / 4, 1, a.ppack, nct_o, wt2, lane, lr, lg, po, wpp);
                    else
                        gemm_stream_plain<2, 16, true>(drop.on ? HD : Atop, drop.on ? 64 : KAP, H, H / 4, 1, a.ppack, nct_o, wt2, lane, lr, lg, po);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (i == 1 && !two) continue;
                        const int c0 = wt2[i] * 16 + 4 * lg, c1 = wt2[i] * 16 + lr;
                        if (c0 < Dout && valid[0]) {                    // Dout % 4 == 0: whole float4 pieces
                            const size_t o = (s * N + lr) * Dout + c0;
                            st4(a.out + o, po[i][0]);
                            if (t + 1 < T) {
                                const f32x4 nxt = tf ? ld4(a.targets + o) : po[i][0];
                                st4(a.xin + o + xstep, nxt);
                                st4(XA + lr * XS + c0, nxt);            // X0 slot 0 of the next step
                            }
                        }
                        if (c1 < Dout && valid1) {                      // node 16 + lg, one column per lane
                            const size_t o = (s * N + node1) * Dout + c1;
                            const float v = po[i][1][0];
                            a.out[o] = v;
                            if (t + 1 < T) {
                                const float nxt = tf ? a.targets[o] : v;
                                a.xin[o + xstep] = nxt;
                                XA[node1 * XS + c1] = nxt;
                            }
                        }
                    }
                }
            }
            __syncthreads();                                         // X0 slot 0 of the next step complete
            pp.mark(7);
        }
    }
    pp.dump(a.probe);
}

}  // namespace eeg

namespace eeg {

// ---- persistent decoder backward ---------------------------------------------------------------------------------
// The BPTT mirror of dec_fwd_persist_kernel: ONE launch walks the T_out steps backwards for the clips it owns -- per
// step the projection transpose and every layer's cell backward (the step of seq_bwd_kernel: blend backward, P^T dC,
// GEMM1, dR, P^T [dR|dU], GEMM2) -- with every weight streamed from L2.  The input gradient of a layer needs no GEMM
// of its own and no adjoint node mix: dX = sum_m P_m^T (dXW W^x_m^T) = sum_m (P_m^T dXW) W^x_m^T, and the adjoint hop
// rows P_m^T dC / P_m^T [dR|dU] are exactly what GEMM1 / GEMM2 consume -- so the packs c1 / c2 (kernels_pack.h) carry
// the input-feature columns next to the hidden ones and the two GEMMs accumulate dX as extra column tiles from the same
// operand fragments.  For a layer above the first, dX (64 columns, tile w on wave w) IS the next cell's external
// gradient, in registers; for the first layer it goes to an LDS tile and, through the autoregressive feedback, into the
// previous step's output gradient.  On chip per clip: the dC / [dR|dU] tiles with their adjoint hop rows, the
// output-gradient tiles, the recurrent gradients dh^l (lane-linear LDS slots).  It emits dXW of every (layer, step) --
// the hoisted parameter-gradient GEMMs, the bias column sums and the projection gradients stay as they are -- the total
// output gradients dOtot and dh0.  At most 20 nodes, 64 units, Dout <= 128 (packs of 8 or 12 column tiles).
struct DecBwdLayerPtrs {
    const float *c1, *c2;                            // weight packs [hidden | input] (kernels_pack.h)
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
    float *dbias0, *dbias1;       // (B,3H) per-clip bias-gradient sums [r|u|c] over steps and nodes: layer 0, layers >= 1 (one shared cell)
    unsigned long long feeds_mask;     // bit t: out_t is the input of step t+1 (no teacher forcing there, t+1 < T)
    const int* teacher_dev;            // nullable DEVICE int32[T] teacher-forcing flags: when set, feeds_mask is derived from it at launch
    int p_batched, T, B, N, Dout, L, act;
    DropCfg drop;                      // dropout in front of the projection (DecFwdArgs): d h_top = mask * (dO W_p), mask recomputed
    const unsigned long long* rng_used;
    long long* probe;                  // phase cycles (development builds with -DEEG_DEC_PROBE), else unused
};

__host__ __device__ constexpr size_t dec_bwd_lds_floats(int M, int L, int Dout) {
    const int H = 64, FS = lds_stride_q(round_up(Dout, 16));
    return (size_t)(M - 1) * kPFloats + (size_t)kDecRows * (M * H + M * 2 * H) + 2 * (size_t)kDecRows * FS
           + (size_t)L * 4 * 2 * 256;
}

template <int H, int M, int DT>
__global__ __launch_bounds__(256, 1) void dec_bwd_persist_kernel(DecBwdArgs a) {
    unsigned long long feeds_mask = a.feeds_mask;
    if (a.teacher_dev != nullptr) {                 // flags drawn on the device: out_t feeds step t+1 unless teacher-forced
        feeds_mask = 0;
        for (int t = 0; t + 1 < a.T; ++t)
            if (a.teacher_dev[t] == 0) feeds_mask |= 1ull << t;
    }
    static_assert(H == 64, "one column tile per wave");
    constexpr int NKS = 5, ROWS = kDecRows, KAP = M * H, KGP = M * 2 * H, NCT = H / 16, NQ = M * H / 16;
    constexpr int PD = NQ < 3 ? NQ : 3;              // quads of weights in flight ahead of the MFMAs
    constexpr int kCx = cell_pack_cx_cols(H, H) / 16;     // column tiles of the c1 / c2 packs (12)
    EEG_DYN_SMEM(sm);
    const int T = a.T, B = a.B, N = a.N, Dout = a.Dout, L = a.L, act = a.act;
    const int FP = round_up(Dout, 16), FS = lds_stride_q(FP);      // row stride of the output-gradient tiles: 4 mod 64 dwords (conflict-free fragment reads)
    float* Pl = sm;
    float* EC = Pl + (M - 1) * kPFloats;     // [ROWS][KAP]  slot 0 = dC, slots m = P_m^T dC
    float* EG = EC + ROWS * KAP;             // [ROWS][KGP]  slot 0 = [dR|dU], slots m = P_m^T [dR|dU]
    float* DO = EG + ROWS * KGP;             // [ROWS][FS]   total gradient of out_t
    float* DX = DO + ROWS * FS;              // [ROWS][FS]   input gradient of layer 0 at step t (feeds dO_{t-1})
    float* DH = DX + ROWS * FS;              // [L][4 waves][2][64 lanes] float4: recurrent gradients dh^l, lane-linear
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), lr = lane & 15, lg = lane >> 4;
    const int ct = wave, col = ct * 16 + 4 * lg;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const size_t state = (size_t)B * N * H;
    const int nct_h = H / 16, nct_o = FP / 16;
    // column tiles of c1 / c2 this wave accumulates: its hidden tile, then the input-feature tiles wave, wave + 4
    const int nx0 = (nct_o + 3) / 4;                                // input tiles per wave, layer 0 (1 or 2)
    const int wt0[3] = {ct, NCT + (wave < nct_o ? wave : 0), NCT + (wave + 4 < nct_o ? wave + 4 : 0)};
    const int wtu[3] = {ct, NCT + ct, NCT + ct};                    // layers above: input = 64 hidden units of the layer below

    DecProbe<kDecProbe> pp;        // 0 output-gradient tile, 1 projection transpose, 2 blend backward + node mix + barrier, 3 GEMM1, 4 its epilogue
    pp.start();                    // + node mixes + barrier, 5 GEMM2, 6 hand-over of dX + barrier, 7 clip set-up / bias sums
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < (int)(ROWS * (KAP + KGP) + 2 * ROWS * FS + L * 4 * 2 * 256); e += 256) EC[e] = 0.f;
        lds_load_polys(Pl, a.P, a.p_batched ? b : 0, M, N);
        __syncthreads();
        float pf[poly_slots<M, NKS>()][NKS];
        load_poly_frags<M, NKS, true>(Pl, pf, lr, lg);
        const int node[2] = {lr, 16 + lr};
        const bool valid[2] = {lr < N, 16 + lr < N};
        const int nodec[2] = {valid[0] ? lr : N - 1, valid[1] ? 16 + lr : N - 1};
        const int oh[2] = {nodec[0] * H + col, nodec[1] * H + col};
        const int oxw[2] = {node[0] * (3 * H) + col, node[1] * (3 * H) + col};
        // remainder nodes 16..19: one element per lane -- lane (lr, lg) <-> node 16 + lg, column ct*16 + lr; every nt = 1
        // quantity below lives in component 0 of its vector
        const int node1 = 16 + lg, col1 = ct * 16 + lr;
        const bool valid1 = node1 < N;
        const int oh1 = (valid1 ? node1 : N - 1) * H + col1, oxw1 = node1 * (3 * H) + col1;
        const int lc1 = lds_sw(node1, col1, KAP), lg1 = lds_sw(node1, col1, KGP), lu1 = lds_sw(node1, H + col1, KGP);
        float* dhl = DH + (wave * 2) * 256 + 4 * lane;                 // + l * 2048 + nt * 256
        const size_t boff = (size_t)b * N * H;
        // operands of one (layer, step) pair; the next pair's are requested while the current one is processed
        f32x4 nh[2], nr[2], nu[2], nc[2];
        auto fetch = [&](int l, int t) {
            const DecBwdLayerPtrs& lp = a.l[l];
            const size_t so = (size_t)t * state + boff;
            nh[0] = ld4(lp.hext + so + oh[0]);                         // hext slot t = h_{t-1}
            nr[0] = ld4(lp.rs + so + oh[0]);
            nu[0] = ld4(lp.us + so + oh[0]);
            nc[0] = ld4(lp.cs + so + oh[0]);
            nh[1] = (f32x4){lp.hext[so + oh1], 0.f, 0.f, 0.f};
            nr[1] = (f32x4){lp.rs[so + oh1], 0.f, 0.f, 0.f};
            nu[1] = (f32x4){lp.us[so + oh1], 0.f, 0.f, 0.f};
            nc[1] = (f32x4){lp.cs[so + oh1], 0.f, 0.f, 0.f};
        };
        // column tiles of a pair: 1 (no input gradient wanted), 2 (layers above the first / narrow outputs) or 3
        auto pair_nt = [&](int l, int t) {
            if (l > 0) return 2;
            return t > 0 && ((feeds_mask >> (t - 1)) & 1ull) != 0 ? 1 + nx0 : 1;
        };
        // weights requested ahead of their GEMM: the first group of the projection transpose, the first PD quads of GEMM1 / GEMM2
        const int wt1[1] = {ct};
        float wpt[DT][1], wq1[PD + 1][4][3], wq2[PD + 1][4][3];
        // (kCx column tiles in every c1 / c2 pack: the weight addresses are base + immediate; narrower pairs skip tiles)
        auto prefetch1 = [&](int l, int nt_) {
            const int(&wt)[3] = l == 0 ? wt0 : wtu;
            const wbuf_t wb = make_wbuf(a.l[l].c1);
            const unsigned v0 = wt[0] * 64 + lane, v1 = wt[1] * 64 + lane, v2 = wt[2] * 64 + lane;
#pragma unroll
            for (int q = 0; q < PD; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) wq1[q][j][0] = wbuf_ld(wb, v0, (4 * q + j) * kCx * 64);
            if (nt_ > 1) {
#pragma unroll
                for (int q = 0; q < PD; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) wq1[q][j][1] = wbuf_ld(wb, v1, (4 * q + j) * kCx * 64);
            }
            if (nt_ > 2) {
#pragma unroll
                for (int q = 0; q < PD; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) wq1[q][j][2] = wbuf_ld(wb, v2, (4 * q + j) * kCx * 64);
            }
        };
        f32x4 sb0[3] = {zero4, zero4, zero4}, sb1[3] = {zero4, zero4, zero4};   // bias-gradient sums [r, u, c]: layer 0, layers above
        float sr0[3] = {0.f, 0.f, 0.f}, sr1[3] = {0.f, 0.f, 0.f};               // ... of the remainder element
        fetch(L - 1, T - 1);
        plain_wload<1, DT>(a.tpack, nct_h, wt1, lane, 0, wpt);
        pp.mark(7);
        for (int t = T - 1; t >= 0; --t) {
            const size_t s = (size_t)t * B + b;
            const bool fb = ((feeds_mask >> t) & 1ull) != 0;       // out_t feeds step t+1: its gradient gets DX of that step
            // ---- S0: total gradient of out_t -> DO tile and dOtot
            for (int e = tid; e < N * (Dout / 4); e += 256) {
                const int n = e / (Dout / 4), c4 = e % (Dout / 4);
                f32x4 g = ld4(a.dOut + (s * N + n) * Dout + 4 * c4);
                if (fb) g += ld4(DX + n * FS + 4 * c4);
                st4(DO + n * FS + 4 * c4, g);
                st4(a.dOtot + (s * N + n) * Dout + 4 * c4, g);
            }
            __syncthreads();
            pp.mark(0);
            // ---- gradient of h^{L-1}_t through the projection (model.py:188-190): dA = dO W_p
            f32x4 gext[2];
            {
                f32x4 pa[1][2] = {{zero4, zero4}};
                gemm_stream_plain<1, DT, false, true>(DO, FS, FP, Dout / 4, 1, a.tpack, nct_h, wt1, lane, lr, lg, pa, wpt);
                gext[0] = pa[0][0];
                gext[1] = pa[0][1];
                if (a.drop.on) {      // through the dropout in front of the projection: the forward's mask, recomputed
                    gext[0] *= dropout_mask4(a.rng_used[0], a.rng_used[1], (s * N * H + oh[0]) >> 2, a.drop.thr, a.drop.scale);
                    gext[1][0] *= dropout_mask1(a.rng_used[0], a.rng_used[1], s * N * H + oh1, a.drop.thr, a.drop.scale);
                }
            }
            prefetch1(L - 1, pair_nt(L - 1, t));
            pp.mark(1);
            for (int l = L - 1; l >= 0; --l) {
                const DecBwdLayerPtrs& lp = a.l[l];
                float* dxw = lp.dxw + s * N * (3 * H);
                const int(&wt)[3] = l == 0 ? wt0 : wtu;
                f32x4 hp[2], rr[2], uu[2], cc[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) { hp[nt] = nh[nt]; rr[nt] = nr[nt]; uu[nt] = nu[nt]; cc[nt] = nc[nt]; }
                if (l > 0) fetch(l - 1, t); else if (t > 0) fetch(L - 1, t - 1);
                // ---- E1: blend backward (cell.py:182-210 reversed)
                f32x4 dU[2], dhn[2];
                {
                    const f32x4 h = hp[0], u = uu[0], c = cc[0];
                    const f32x4 g = valid[0] ? ld4(dhl + l * 2048) + gext[0] : zero4;
                    f32x4 dC, du_;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dc = g[r] * (1.f - u[r]);
                        dC[r] = act == 0 ? dc * (1.f - c[r] * c[r]) : (c[r] > 0.f ? dc : 0.f);
                        du_[r] = g[r] * (h[r] - c[r]) * u[r] * (1.f - u[r]);
                    }
                    st4(EC + lds_sw(lr, col, KAP), dC);                               // zeros on padding nodes
                    if (valid[0]) {
                        st4(dxw + oxw[0] + 2 * H, dC);
                        st4(dxw + oxw[0] + H, du_);
                    }
                    dU[0] = du_;
                    dhn[0] = g * u;
                    if (l == 0) { sb0[1] += du_; sb0[2] += dC; } else { sb1[1] += du_; sb1[2] += dC; }
                    // node 16 + lg, column ct*16 + lr
                    const float h1 = hp[1][0], u1 = uu[1][0], c1 = cc[1][0];
                    const float g1 = valid1 ? dhl[l * 2048 + 256] + gext[1][0] : 0.f;
                    const float dc1 = g1 * (1.f - u1);
                    const float dC1 = act == 0 ? dc1 * (1.f - c1 * c1) : (c1 > 0.f ? dc1 : 0.f);
                    const float du1 = g1 * (h1 - c1) * u1 * (1.f - u1);
                    EC[lc1] = dC1;
                    if (valid1) {
                        dxw[oxw1 + 2 * H] = dC1;
                        dxw[oxw1 + H] = du1;
                    }
                    dU[1] = (f32x4){du1, 0.f, 0.f, 0.f};
                    dhn[1] = (f32x4){g1 * u1, 0.f, 0.f, 0.f};
                    if (l == 0) { sr0[1] += du1; sr0[2] += dC1; } else { sr1[1] += du1; sr1[2] += dC1; }
                }
                EEG_WAVE_SYNC();
                lds_diffuse_tile<M, NKS, ROWS>(EC, KAP, ct * 16, H, pf, lr, lg);
                __syncthreads();                                         // (1) P_m^T dC complete
                pp.mark(2, l);
                const int ntp = pair_nt(l, t);
                f32x4 dx[2][2] = {{zero4, zero4}, {zero4, zero4}};      // this wave's input-gradient tiles
                auto cell = [&](auto ntag) {
                    constexpr int NT = decltype(ntag)::value, nct = kCx;
                    int wtn[NT];
#pragma unroll
                    for (int i = 0; i < NT; ++i) wtn[i] = wt[i];
                    // ---- GEMM1: [d(r*h) | dX part] = [P_m^T dC]_m @ [Wc^h | Wc^x]^T
                    f32x4 acc[NT][2];
#pragma unroll
                    for (int i = 0; i < NT; ++i) { acc[i][0] = zero4; acc[i][1] = zero4; }
                    gemm_stream_quad<NT, NQ, PD, true, 3>(EC, KAP, lp.c1, nct, wtn, lane, lr, lg, acc, wq1);
                    quad_prefetch<NT, 2 * NQ, PD, 3>(lp.c2, nct, wtn, lane, wq2);
                    pp.mark(3, l);
                    {
                        const f32x4 drh = acc[0][0], rg = rr[0];         // exact 0 on padding nodes
                        const f32x4 dR = drh * hp[0] * rg * (1.f - rg);
                        dhn[0] += drh * rg;
                        st4(EG + lds_sw(lr, col, KGP), dR);
                        st4(EG + lds_sw(lr, H + col, KGP), dU[0]);
                        if (valid[0]) st4(dxw + oxw[0], dR);
                        if (l == 0) sb0[0] += dR; else sb1[0] += dR;
                        acc[0][0] = dhn[0];
                        const float drh1 = acc[0][1][0], rg1 = rr[1][0];
                        const float dR1 = drh1 * hp[1][0] * rg1 * (1.f - rg1);
                        dhn[1][0] += drh1 * rg1;
                        EG[lg1] = dR1;
                        EG[lu1] = dU[1][0];
                        if (valid1) dxw[oxw1] = dR1;
                        if (l == 0) sr0[0] += dR1; else sr1[0] += dR1;
                        acc[0][1] = (f32x4){dhn[1][0], 0.f, 0.f, 0.f};
                    }
                    EEG_WAVE_SYNC();
                    lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, ct * 16, 2 * H, pf, lr, lg);
                    lds_diffuse_tile<M, NKS, ROWS>(EG, KGP, H + ct * 16, 2 * H, pf, lr, lg);
                    __syncthreads();                                     // (2) P_m^T [dR|dU] complete
                    pp.mark(4, l);
                    // ---- GEMM2: [dh | dX] += [P_m^T dG]_m @ [Wg^h | Wg^x]^T -> recurrent gradient for step t-1, input gradient
                    gemm_stream_quad<NT, 2 * NQ, PD, true, 3>(EG, KGP, lp.c2, nct, wtn, lane, lr, lg, acc, wq2);
                    pp.mark(5, l);
                    st4(dhl + l * 2048 + 0 * 256, acc[0][0]);
                    st4(dhl + l * 2048 + 1 * 256, acc[0][1]);
#pragma unroll
                    for (int i = 1; i < NT; ++i) { dx[i - 1][0] = acc[i][0]; dx[i - 1][1] = acc[i][1]; }
                };
                if (ntp == 3) cell(std::integral_constant<int, 3>{});
                else if (ntp == 2) cell(std::integral_constant<int, 2>{});
                else cell(std::integral_constant<int, 1>{});
                // the next pair's first weights fly across the barrier
                if (l > 0) prefetch1(l - 1, pair_nt(l - 1, t));
                else if (t > 0) plain_wload<1, DT>(a.tpack, nct_h, wt1, lane, 0, wpt);
                if (l > 0) {
                    gext[0] = dx[0][0];                                  // gradient of h^{l-1}_t, already in this wave's registers
                    gext[1] = dx[0][1];
                } else if (ntp > 1) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int j = wave + 4 * i;
                        if (j >= nct_o || i + 1 >= ntp) continue;
                        st4(DX + lr * FS + j * 16 + 4 * lg, dx[i][0]);
                        DX[node1 * FS + j * 16 + lr] = dx[i][1][0];
                    }
                }
                __syncthreads();                                         // (3) tiles free for the next pair; DX complete
                pp.mark(6, l);
            }
        }
        // ---- gradients of the initial states (the encoder's final states)
        if (a.dh0 != nullptr) {
            for (int l = 0; l < L; ++l) {
                if (valid[0]) st4(a.dh0 + (size_t)l * state + boff + lr * H + col, ld4(dhl + l * 2048));
                if (valid1) a.dh0[(size_t)l * state + boff + node1 * H + col1] = dhl[l * 2048 + 256];
            }
        }
        // ---- per-clip bias-gradient sums (fixed-order node reduction through the free dC / [dR|dU] tiles)
        float* red = EC;                                             // [3H][16] + [3H][4] (the remainder elements)
        float* red1 = EC + 3 * H * 16;
        for (int set = 0; set < (L > 1 ? 2 : 1); ++set) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(k * H + col + r) * 16 + lr] = set == 0 ? sb0[k][r] : sb1[k][r];
                red1[(k * H + col1) * 4 + lg] = set == 0 ? sr0[k] : sr1[k];
            }
            __syncthreads();
            float* dst = (set == 0 ? a.dbias0 : a.dbias1) + (size_t)b * 3 * H;
            for (int j = tid; j < 3 * H; j += 256) {
                float sacc = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) sacc += red[j * 16 + q];
                sacc += (red1[j * 4] + red1[j * 4 + 1]) + (red1[j * 4 + 2] + red1[j * 4 + 3]);
                dst[j] = sacc;
            }
        }
        pp.mark(7);
    }
    pp.dump(a.probe);
}

}  // namespace eeg
