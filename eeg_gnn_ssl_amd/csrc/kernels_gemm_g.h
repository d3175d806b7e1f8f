// Grouped hoisted GEMMs of the spectral form (kernels_spectral.h): one plain GEMM per graph frequency i, all in ONE launch.
//
//  gemm_nng_kernel: C[i*Sp + r][:] = A_i[r][0:F) * W_i for the G row groups of Sp rows (Sp % 16 == 0) of a node-major
//  (G, Sp, F) tensor (group i of A starts a_gstride floats behind group i-1: the transformed input may be a row window of the
//  layer below's (N, SpE, H) by-product) -- gemm_nnr_kernel (kernels_gemm_q.h: persistent balanced row ranges, LDS ring filled by
//  buffer_load ... lds, ds_read_b128 fragments against quad-ordered weights streamed from L2 into registers, transposed MFMA
//  issue) with the right-hand side selected per 128-row tile.  Differences:
//   * one A segment (nseg = 1); a tile never straddles a group: the tile walk of a workgroup's row range is cut at the group
//     boundaries (tiles of 1..8 row tiles), and two cursors walk it -- the request cursor three chunks ahead and the MFMA cursor;
//   * the weights of a chunk arrive by LDS-DMA next to the activations (each wave requests and reads only its own column tiles), so
//     the time loop has no load that returns into registers and every wait on the vector-memory queue is a COUNTED one written
//     here: with register-returning weight loads in the loop the compiler put a drain (vmcnt(0)) behind every chunk -- the requests
//     issued a moment earlier included -- and a tile's C stores had to retire within one chunk;
//   * every row tile is whole (Sp % 16 == 0) and every column block is whole (nct == 4 * NJ): no guards on the stores;
//   * NJ column tiles per wave: 3 (192 columns: the pre-activations [r|u|c] of a 64-unit cell) or 1 (64 columns: dX of a layer
//     above the first); the cell's bias enters as the accumulator start gscale[group] * bias[col] (a node-constant row b is
//     (sum_n U[n][i]) * b in the eigenbasis: the node mix behind this GEMM then needs no bias).
//  K order: the F / 16 whole chunks, then one tail chunk with the (F / 4) % 4 left-over pieces (zero weights behind them; the
//  lanes of the padding pieces fetch columns 0..3 of their row: finite values times zero).
// Reference semantics: model/cell.py:98-117 in the eigenbasis of the support.
#pragma once
#include "kernels_gemm_q.h"

namespace eeg {

#ifndef EEG_X_NNG_STAUX
#define EEG_X_NNG_STAUX 0
#endif
constexpr int kNngStages = 4;
// LDS of one workgroup: the ring of kNngStages stages, each 128 rows x 16 floats of activations + 4 NJ column tiles of weights
__host__ __device__ constexpr size_t nng_lds_bytes(int NJ) { return (size_t)kNngStages * (128 * 16 + 4 * NJ * 256) * sizeof(float); }

template <int NJ, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_nng_kernel(const float* __restrict__ A, unsigned a_gstride, int F, int Sp, int G,
                                                         const float* __restrict__ Wq, unsigned w_group_stride,
                                                         float* __restrict__ C, int ldc,
                                                         const float* __restrict__ bias, const float* __restrict__ gscale) {
    constexpr int NS = kNngStages, STA = 128 * 16, NCT = 4 * NJ, ST = STA + NCT * 256, NPER = 2 + NJ;
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), lr = lane & 15, lg = lane >> 4;
    const NnqOrder ko = make_nnq_order(1, F);
    const int nch = ko.nch;
    const int RTg = Sp / 16, RT = RTg * G, Rtot = RT * 16;
    const int Gd = gridDim.x, bid = blockIdx.x;
    const int rt0 = (int)((long long)bid * RT / Gd), rt1 = (int)((long long)(bid + 1) * RT / Gd);
    if (rt1 <= rt0) return;
    // tile walk: a tile starts at `cur` and ends at the next of {cur + 8, rt1, group boundary}
    auto tile_len = [&](int cur) __attribute__((always_inline)) -> int {
        int len = rt1 - cur;
        if (len > 8) len = 8;
        const int to_boundary = RTg - cur % RTg;
        return len < to_boundary ? len : to_boundary;
    };
    int ntile = 0;
    for (int cur = rt0; cur < rt1; cur += tile_len(cur)) ++ntile;
    const int Q = ntile * nch;

    // ---- DMA side: per chunk and wave two requests of activations (row tiles w, w + 4) and NJ of weights (its own column tiles) ------
    const wbuf_t ra = make_wbuf(A);
    const wbuf_t rb = make_wbuf(Wq + (size_t)(NJ * w) * 256);
    const int a_piece = (lane & 3) ^ nnq_gsw(lg);
    // tail chunk: piece pp is valid for pp < ko.b (columns 16a + 4pp ..), else columns 0..3 (their weights are zero)
    const int t_adj = a_piece < ko.b ? 4 * (ko.a * 16) : -16 * a_piece;       // bytes, relative to the main-chunk lane offset
    int d_cur = rt0, d_c = 0, d_stage = 0;
    unsigned a_voff[2], d_wbase = (unsigned)(rt0 / RTg) * w_group_stride;   // (floats: the weight block of the request cursor's group)
    auto tile_rows = [&](int cur) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r = cur * 16 + 16 * (w + 4 * i) + (lane >> 2);
            if (r >= Rtot) r = Rtot - 1;
            const int g = r / Sp;                                             // (rows behind a short tile may belong to the next group)
            a_voff[i] = ((unsigned)g * a_gstride + (unsigned)(r - g * Sp) * F + 4 * a_piece) * 4u;
        }
    };
    tile_rows(d_cur);
    auto issue = [&]() __attribute__((always_inline)) {
        float* base = sm + d_stage * ST;
        if (d_c < ko.nmain) {
            wbuf_dma16(ra, base + w * 256, a_voff[0], (unsigned)d_c * 64u);
            wbuf_dma16(ra, base + (w + 4) * 256, a_voff[1], (unsigned)d_c * 64u);
        } else {
            wbuf_dma16(ra, base + w * 256, a_voff[0] + (unsigned)t_adj, 0u);
            wbuf_dma16(ra, base + (w + 4) * 256, a_voff[1] + (unsigned)t_adj, 0u);
        }
        const unsigned wso = (d_wbase + (unsigned)(d_c * NCT) * 256u) * 4u;
#pragma unroll
        for (int j = 0; j < NJ; ++j) wbuf_dma16(rb, base + STA + (NJ * w + j) * 256, (unsigned)lane * 16u, wso + (unsigned)j * 1024u);
        d_stage = d_stage + 1 == NS ? 0 : d_stage + 1;
        if (++d_c == nch) {
            d_c = 0;
            d_cur += tile_len(d_cur);
            if (d_cur < rt1) {
                tile_rows(d_cur);
                d_wbase = (unsigned)(d_cur / RTg) * w_group_stride;
            }
        }
    };

    // ---- compute side ----------------------------------------------------------------------------------------------------
    const int c_col = 16 * NJ * w + 4 * lg;
    const wbuf_t rc = make_wbuf(C);
    const int a_lds = lr * 16 + 4 * (lg ^ nnq_gsw((lr >> 2) & 3));
    const int w_lds = STA + (NJ * w) * 256 + lane * 4;
    f32x4 acc[8][NJ], oa[8], ob[NJ], obn[NJ];
    // bias (nullable): the tiles of group g start from gscale[g] * bias[col] -- a per-node-constant bias row in the eigenbasis
    f32x4 bia[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bia[j] = bias != nullptr ? *reinterpret_cast<const f32x4*>(bias + c_col + 16 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
    auto acc_init = [&](int cur) __attribute__((always_inline)) {
        const float sc = bias != nullptr ? gscale[wave_uniform(cur / RTg)] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = bia[j] * sc;
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < Q) issue();
    acc_init(rt0);
    __syncthreads();                                                         // the prologue requests of all waves (drains the queue)
#pragma unroll
    for (int i = 0; i < 8; ++i) oa[i] = *reinterpret_cast<const f32x4*>(sm + a_lds + i * 256);
#pragma unroll
    for (int j = 0; j < NJ; ++j) ob[j] = *reinterpret_cast<const f32x4*>(sm + w_lds + j * 256);
    int r_stage = 1, m_c = 0, m_cur = rt0, nrt = tile_len(rt0);
    bool burst = false;                                                      // the previous chunk ended a full tile: its 8 * NJ stores are in the queue
    // One chunk.  The loop has no vector-memory instruction that returns into registers (activations AND weights arrive by LDS-DMA),
    // so the only waits on the queue are the counted ones below: at the end of chunk q the requests of chunk q+2 (issued a chunk ago)
    // must have landed; younger than them in the queue (it retires in order, stores included: tools/micro/order_lab) are the
    // requests of chunk q+3 and -- behind a tile end -- that tile's C stores, which thereby get two chunk times to drain.
    auto chunk = [&](auto DMA_, auto MORE_) __attribute__((always_inline)) {
        constexpr bool dma = decltype(DMA_)::value, more = decltype(MORE_)::value;
        if (more) EEG_LDS_BARRIER();                                         // chunk q+1 landed in every wave
#ifndef EEG_X_NNG_NODMA
        if (dma) issue();                                                    // chunk q+3 into the stage of chunk q-1
#endif
        EEG_SCHED_FENCE();
        const float* st = sm + r_stage * ST;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < nrt) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16(ob[j][s], oa[i][s], acc[i][j]);   // transposed issue
            }
            oa[i] = *reinterpret_cast<const f32x4*>(st + a_lds + i * 256);  // refilled in place from chunk q+1
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) obn[j] = *reinterpret_cast<const f32x4*>(st + w_lds + j * 256);
        EEG_SCHED_FENCE();
        if (more) {
            if (!dma) EEG_VM_WAIT(0);
            else if (burst) vm_wait_n<NPER + 8 * NJ>();
            else vm_wait_n<NPER>();
#pragma unroll
            for (int j = 0; j < NJ; ++j) ob[j] = obn[j];
        }
        burst = false;
        r_stage = r_stage + 1 == NS ? 0 : r_stage + 1;
        if (++m_c == nch) {                                                  // the tile of chunk q is complete
            const int row0 = m_cur * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#ifndef EEG_X_NNG_NOSTORE
                    if (i < nrt) wbuf_st4_aux<EEG_X_NNG_STAUX>(rc, (unsigned)(lr * ldc + c_col + 16 * j), (unsigned)(row0 + 16 * i) * (unsigned)ldc, acc[i][j]);
#else
                    if (i < nrt && acc[i][j][0] == 1.2345e-33f) wbuf_st4(rc, (unsigned)(lr * ldc + c_col + 16 * j), (unsigned)(row0 + 16 * i) * (unsigned)ldc, acc[i][j]);
#endif
            EEG_SCHED_FENCE();                                               // (a store's data registers must not be rewritten right behind it)
#ifndef EEG_X_NNG_NOSTORE
            burst = nrt == 8;
#endif
            m_c = 0;
            m_cur += nrt;
            nrt = m_cur < rt1 ? tile_len(m_cur) : 0;
            acc_init(m_cur < rt1 ? m_cur : rt0);
        }
    };
    int q = 0;
    for (; q + NS - 1 < Q; ++q) chunk(IntC<1>(), IntC<1>());
    for (; q + 1 < Q; ++q) chunk(IntC<0>(), IntC<1>());
    if (q < Q) chunk(IntC<0>(), IntC<0>());
}

// Grouped weight-gradient GEMM: gemm_tnq_body (kernels_gemm_q.h) with the row range of a workgroup taken from ONE group:
// split y = i * spg + ls covers rows [i*Sp + ls*rps, min(i*Sp + (ls+1)*rps, (i+1)*Sp)) and writes partial[y][K][Ov] =
// A_i^T dY_i[:, ycol0 : ycol0 + Ov].
template <int KT, int OT, int RC, bool PLANAR>
__global__ __launch_bounds__(256, 2) void gemm_tnq_grouped_kernel(SegPtrs segs, int F, int Sp, int G, int spg,
                                                                 const float* __restrict__ dY, int ldy, int ycol0, int Ov,
                                                                 float* __restrict__ partial, int rows_per_split, long long a_gskew) {
    const int y = (int)blockIdx.y, i = y / spg, ls = y - i * spg;
    const int rbeg = i * Sp + ls * rows_per_split;
    int rend = rbeg + rows_per_split;
    if (rend > (i + 1) * Sp) rend = (i + 1) * Sp;
    // a_gskew = (floats between two groups of A) - Sp * F: all rows of this workgroup lie in group i, so shifting the base makes
    // the linear row index i*Sp + r address row r of group i
    segs.p[0] += (long long)i * a_gskew;
    gemm_tnq_rows<KT, OT, RC, false, PLANAR, false>(segs, 1, F, Sp * G, dY, ldy, ycol0, Ov, partial, 0, 0, 0, 0, (int)blockIdx.x, y, rbeg, rend);
}
// The two h-part problems of a cell in the eigenbasis -- (U^T h)_i^T dYh_i[:, 0:2H] and (U^T (r*h))_i^T dYh_i[:, 2H:3H] -- as ONE launch
// whose workgroups alternate between the two over the workgroup slots of a CU (cf. gemm_tnq_pair_kernel).
template <int KT, int RC, bool PLANAR>
__global__ __launch_bounds__(256, 2) void gemm_tnq_grouped_pair_kernel(TnqJob ja, TnqJob jb, int F, int Sp, int G, int spg,
                                                                      const float* __restrict__ dY, int ldy, int rows_per_split,
                                                                      long long a_gskew, long long b_gskew) {
    const int yy = (int)blockIdx.y, y = yy >> 1, i = y / spg, ls = y - i * spg;
    ja.segs.p[0] += (long long)i * a_gskew;
    jb.segs.p[0] += (long long)i * b_gskew;
    const int which = (yy & 1) ^ ((yy >> 8) & 1);
    const int rbeg = i * Sp + ls * rows_per_split;
    int rend = rbeg + rows_per_split;
    if (rend > (i + 1) * Sp) rend = (i + 1) * Sp;
    if (which == 0)
        gemm_tnq_rows<KT, 4, RC, false, PLANAR, false>(ja.segs, 1, F, Sp * G, dY, ldy, ja.ycol0, ja.Ov, ja.partial, 0, 0, 0, 0, (int)blockIdx.x, y, rbeg, rend);
    else
        gemm_tnq_rows<KT, 2, RC, false, PLANAR, false>(jb.segs, 1, F, Sp * G, dY, ldy, jb.ycol0, jb.Ov, jb.partial, 0, 0, 0, 0, (int)blockIdx.x, y, rbeg, rend);
}

}  // namespace eeg
