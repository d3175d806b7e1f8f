// Lab-only NN GEMM kernels of round 3 (tools/micro/gemm_lab.hip): gemm_nnq_kernel = the first version of the persistent
// quad-packed design (weights staged in LDS beside the activations; carries the ablation switches the lab reports) and
// gemm_nnl_kernel = the same with a loader wave (measured slower).  The shipped kernel is gemm_nnr_kernel in
// eeg_gnn_ssl_amd/csrc/kernels_gemm_q.h, whose header describes the design they share.
#pragma once
#include "../../eeg_gnn_ssl_amd/csrc/kernels_gemm_q.h"

namespace eeg {

constexpr int kNnqStageFloats = 128 * 16 + 12 * 256;   // A tile + 12 column tiles of the quad pack = 20 KB

// ABL (lab only, bits): 1 = no C stores, 2 = no DMA after the prologue, 4 = A always fetched from the first rows / chunk
// (cache-hot), 8 = B always fetched from chunk 0 (cache-hot), 16 = the DMAs of a chunk are issued between the row tiles
// instead of all at the top, 32 = every tile is stored over the workgroup's first tile (cache-resident C), 64 = the
// stores of a tile interleaved with the MFMAs of its last chunk instead of one burst behind it.  flags bits 0-1: which workgroups start with a HALF first tile (0 none, 1 the upper half of
// the grid, 2 odd ids, 3 bit 3 of the id) -- the two workgroups of a CU then store their tiles half a tile apart.
// Requires: O % 4 == 0, ldc % 4 == 0, F % 4 == 0, at most 2 tail chunks (make_nnq_order(nseg, F).ntail <= 2), every
// segment and C smaller than 4 GB (32-bit buffer offsets).  LDS: NS stages of 20 KB.
template <int NS, int ABL>
__global__ __launch_bounds__(256, 2) void gemm_nnq_kernel(SegPtrs segs, int nseg, int F, int R,
                                                         const float* __restrict__ Bq, int nct_total,
                                                         const float* __restrict__ bias, float* __restrict__ C, int ldc, int O,
                                                         int btT, int btB, int btN, int flags, long long* __restrict__ probe = nullptr) {
    constexpr int NB = 12, AF = 128 * 16, ST = kNnqStageFloats, NST = 24;
    constexpr bool PROBE = (ABL & 128) != 0;   // lab: cycle counters per workgroup (wave 0): probe[8]
    const long long tr0 = PROBE ? realtime_now() : 0;
    long long pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0, pc4 = 0, pc5 = 0, pc6 = 0, pc7 = 0;   // waits after an epilogue (0, 1, 2 iterations), other waits, their count, iteration cycles, epilogue cycles, epilogues
    static_assert(NS >= 2 && NS <= 5, "ring depth");
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), lr = lane & 15, lg = lane >> 4;
    const NnqOrder ko = make_nnq_order(nseg, F);
    const int nch = ko.nch;
    const int RT = ceil_div(R, 16);
    const int rt0 = (int)((long long)blockIdx.x * RT / gridDim.x), rt1 = (int)((long long)(blockIdx.x + 1) * RT / gridDim.x);
    const int nrows = rt1 - rt0;                           // row tiles of this workgroup
    if (nrows <= 0) return;
    const int bid = blockIdx.x, pm = flags & 7;
    const bool half_first = pm == 1 ? bid >= (int)gridDim.x / 2 : pm == 2 ? (bid & 1) : pm == 3 ? ((bid >> 3) & 1) : false;
    int nrt_first = half_first ? 4 : 8;
    if (pm == 4) nrt_first = 4 + bid % 5;                  // five phases: the C stores of the grid spread over the tile period
    if (pm == 5) nrt_first = 1 + bid % 8;
    if (pm == 6) nrt_first = 4 + (bid >> 3) % 5;
    if (nrt_first > nrows) nrt_first = nrows;
    const int ntile = 1 + ceil_div(nrows - nrt_first, 8);
    // tile t covers row tiles [tile_rt(t), tile_rt(t) + tile_nrt(t))
    auto tile_rt = [&](int t) __attribute__((always_inline)) { return rt0 + (t == 0 ? 0 : nrt_first + 8 * (t - 1)); };
    auto tile_nrt = [&](int t) __attribute__((always_inline)) {
        if (t == 0) return nrt_first;
        const int left = nrows - nrt_first - 8 * (t - 1);
        return left < 8 ? left : 8;
    };
    const int ct0 = blockIdx.y * NB;
    const int Q = ntile * nch;

    // ---- DMA side -------------------------------------------------------------------------------------------------
    const wbuf_t rb = make_wbuf(Bq);
    unsigned b_voff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ct = ct0 + w + 4 * i < nct_total ? ct0 + w + 4 * i : nct_total - 1;
        b_voff[i] = (unsigned)(ct * 256 + lane * 4) * 4u;
    }
    const int a_piece = (lane & 3) ^ nnq_gsw(lg);          // logical 16-byte piece this lane fetches (rows 16j + lane/4)
    // tail chunks (leftover 16-byte pieces of all planes): this lane's plane + column, fixed for the launch
    const float* tptr[2];
#pragma unroll
    for (int tc = 0; tc < 2; ++tc) {
        const int tp = tc * 4 + a_piece;
        int seg = 0, f = 0;
        if (tp < nseg * ko.b) { seg = tp / ko.b; f = ko.a * 16 + (tp - seg * ko.b) * 4; }
        tptr[tc] = segs.p[seg] + f;
    }
    int d_tile = 0, d_c = 0, d_seg = 0, d_kc = 0, d_stage = 0;
    unsigned a_voff[2];                                    // (row * F + 4 * piece) * 4 bytes of the two A rows this lane fetches
    auto tile_rows = [&](int tile) __attribute__((always_inline)) {
        const int row0 = ((ABL & 4) ? rt0 : tile_rt(tile)) * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r = row0 + 16 * (w + 4 * i) + (lane >> 2);
            if (r >= R) r = R - 1;
            if (btT > 0) {                                 // batch-major segments (see gemm_nn_dma_kernel)
                const int sm_ = r / btN, n = r - sm_ * btN, t = sm_ / btB, b = sm_ - t * btB;
                r = (b * btT + t) * btN + n;
            }
            a_voff[i] = ((unsigned)r * F + 4 * a_piece) * 4u;
        }
    };
    tile_rows(0);
    // the 5 DMAs of a chunk: part 0 = the two A pieces of this wave, parts 1..3 = its three weight column tiles (and the
    // cursor advance with part 3)
    auto issue_part = [&](int part) __attribute__((always_inline)) {
        float* base = sm + d_stage * ST;
        if (part == 0) {
            if (d_c < ko.nmain) {
                const wbuf_t ra = make_wbuf(segs.p[d_seg]);
                const unsigned so = (ABL & 4) ? 0u : (unsigned)d_kc * 4u;
                wbuf_dma16(ra, base + w * 256, a_voff[0], so);
                wbuf_dma16(ra, base + (w + 4) * 256, a_voff[1], so);
                d_kc += 16;
                if (d_kc == ko.a * 16) { d_kc = 0; ++d_seg; }
            } else {
                const char* p = reinterpret_cast<const char*>(d_c == ko.nmain ? tptr[0] : tptr[1]) - 16 * a_piece;
                lds_dma16(base + w * 256, reinterpret_cast<const float*>(p + a_voff[0]));
                lds_dma16(base + (w + 4) * 256, reinterpret_cast<const float*>(p + a_voff[1]));
            }
            return;
        }
        const unsigned bso = (ABL & 8) ? 0u : (unsigned)(d_c * nct_total) * 1024u;
        wbuf_dma16(rb, base + AF + (w + 4 * (part - 1)) * 256, b_voff[part - 1], bso);
        if (part == 3) {
            d_stage = d_stage + 1 == NS ? 0 : d_stage + 1;
            if (++d_c == nch) {
                d_c = 0; d_seg = 0; d_kc = 0;
                if (++d_tile < ntile) tile_rows(d_tile);
            }
        }
    };
    auto issue_dma = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < 4; ++part) issue_part(part);
    };

    // ---- compute side ---------------------------------------------------------------------------------------------
    // Iteration q multiplies chunk q from registers and, row tile by row tile, refills the activation fragments it has
    // just used from chunk q+1; the 3 weight fragments of chunk q+1 are read at the top and swapped in at the end.
    const int c_col = 16 * (ct0 + 3 * w) + 4 * lg;         // first of this lane's 3 x 4 output columns (+ 16 j)
    const bool cols_full = 16 * (ct0 + 3 * w + 3) <= O;
    const wbuf_t rc = make_wbuf(C);
    const int a_lds = lr * 16 + 4 * (lg ^ nnq_gsw((lr >> 2) & 3));
    const int b_lds = AF + 3 * w * 256 + lane * 4;
    f32x4 acc[8][3], oa[8], ob[3], obn[3];
    int r_stage = 0, m_c = 0, m_tile = 0, epi_age = 100, epi_cnt = 0;
    // the two workgroups of a CU (ids b and b + G/2) share each SIMD's matrix pipe; at equal priority the older one wins
    // every arbitration and finishes far ahead of the other, which then runs alone: alternate who has priority
    const int prio_mode = (flags >> 3) & 3, prio_phase = bid >= (int)gridDim.x / 2 ? 1 : 0;
    if (prio_mode == 3 && prio_phase) EEG_SETPRIO(1);

#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < Q) issue_dma();
    f32x4 bv[3];                                           // accumulators start from the bias
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr && c_col + 16 * j + 3 < O) bv[j] = *reinterpret_cast<const f32x4*>(bias + c_col + 16 * j);
    }
    __syncthreads();                                       // drains the prologue DMAs and the bias loads (once per workgroup)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = bv[j];
#pragma unroll
    for (int i = 0; i < 8; ++i) oa[i] = *reinterpret_cast<const f32x4*>(sm + a_lds + i * 256);
#pragma unroll
    for (int j = 0; j < 3; ++j) ob[j] = *reinterpret_cast<const f32x4*>(sm + b_lds + j * 256);
    r_stage = NS > 1 ? 1 : 0;

    for (int q = 0; q < Q; ++q) {
        const bool more = q + 1 < Q;
        const long long t0 = PROBE ? cycle_now() : 0;
        if (prio_mode == 1) { if ((q + prio_phase) & 1) EEG_SETPRIO(1); else EEG_SETPRIO(0); }
        if (prio_mode == 2 && m_c == 0) { if ((m_tile + prio_phase) & 1) EEG_SETPRIO(1); else EEG_SETPRIO(0); }
        // chunk q+1 must have landed (the DMAs of chunks q+2 .. q+NS-2 and the C stores issued since it was requested may
        // stay in flight); after the barrier every wave has finished reading chunk q-1, whose stage is refilled next
        if (more) {
            if ((ABL & 2) != 0) {
                EEG_LDS_BARRIER();
            } else if (Q - 2 - q >= NS - 3) {
                const bool st = epi_age <= NS - 3 && epi_cnt == NST;
                if (NS == 3) { if (st) EEG_VM_WAIT_BARRIER(24); else EEG_VM_WAIT_BARRIER(0); }
                if (NS == 4) { if (st) EEG_VM_WAIT_BARRIER(29); else EEG_VM_WAIT_BARRIER(5); }
                if (NS == 5) { if (st) EEG_VM_WAIT_BARRIER(34); else EEG_VM_WAIT_BARRIER(10); }
            } else {
                EEG_VM_WAIT_BARRIER(0);
            }
            if (PROBE) {
                const long long dt = cycle_now() - t0;
                if (epi_age == 0) pc0 += dt; else if (epi_age == 1) pc1 += dt; else if (epi_age == 2) pc2 += dt; else { pc3 += dt; pc4 += 1; }
            }
            if ((ABL & (2 | 16)) == 0 && q + NS - 1 < Q) issue_dma();
        }
        const bool spread = (ABL & 16) != 0 && (ABL & 2) == 0 && more && q + NS - 1 < Q;
        // (the last chunk of the range re-reads a stale stage into registers nobody uses: no branch around the reads)
        const float* st = sm + r_stage * ST;
#pragma unroll
        for (int j = 0; j < 3; ++j) obn[j] = *reinterpret_cast<const f32x4*>(st + b_lds + j * 256);
        const int nrt = tile_nrt(m_tile);
        // A CU retires ~17 B/clk of stores: the 96 KB of a tile keep the 4 waves off the matrix pipe for ~5.5 k cycles
        // (probe in tools/micro/gemm_lab.hip).  ABL 64 (lab) issues the 3 stores of a row tile right behind the MFMAs of
        // the next one in the last chunk, so that they drain under MFMAs: the storing workgroup then loses nothing, but
        // the OTHER workgroup of the CU (the younger one: every arbitration goes to the older wave) no longer gets the
        // burst as its turn, falls ~25 % behind and runs alone at the end -- slower overall (0.705 vs 0.738), not shipped.
        const int row0 = ((ABL & 32) ? rt0 : tile_rt(m_tile)) * 16;
        const bool fast_tile = nrt == 8 && row0 + 128 <= R && cols_full;   // exactly NST unconditional stores: the counted waits rely on it
        const bool store_now = (ABL & 64) != 0 && (ABL & 1) == 0 && fast_tile && m_c + 1 == nch;   // lab variant, see below
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < nrt) {                                 // (a partial tile multiplies its own row tiles only)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(ob[j][s], oa[i][s], acc[i][j]);   // transposed issue
            }
            oa[i] = *reinterpret_cast<const f32x4*>(st + a_lds + i * 256);   // refilled in place from chunk q+1
            if ((ABL & 16) != 0 && i < 4 && spread) issue_part(i);
            if (store_now && i > 0) {                      // row tile i-1: its MFMAs have left the pipe by now
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    wbuf_st4(rc, (unsigned)(lr * ldc + c_col + 16 * j), (unsigned)(row0 + 16 * (i - 1)) * (unsigned)ldc, acc[i - 1][j]);
                }
            }
        }
        if (store_now) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                wbuf_st4(rc, (unsigned)(lr * ldc + c_col + 16 * j), (unsigned)(row0 + 16 * 7) * (unsigned)ldc, acc[7][j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) ob[j] = obn[j];
        r_stage = r_stage + 1 == NS ? 0 : r_stage + 1;
        ++epi_age;
        const long long t2 = PROBE ? cycle_now() : 0;
        if (PROBE) pc5 += t2 - t0;
        if (++m_c == nch) {                                // the tile of chunk q is complete
            if (ABL & 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        EEG_USE(acc[i][j]);
                        acc[i][j] = bv[j];
                    }
                epi_cnt = 0;
            } else if (store_now) {                        // stored row tile by row tile above
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = bv[j];
                epi_cnt = NST;
            } else if (fast_tile) {                        // the whole tile in one burst
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        wbuf_st4(rc, (unsigned)(lr * ldc + c_col + 16 * j), (unsigned)(row0 + 16 * i) * (unsigned)ldc, acc[i][j]);
                        acc[i][j] = bv[j];
                    }
                epi_cnt = NST;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        if (i < nrt && row0 + 16 * i + lr < R && c_col + 16 * j < O)
                            wbuf_st4(rc, (unsigned)(lr * ldc + c_col + 16 * j), (unsigned)(row0 + 16 * i) * (unsigned)ldc, acc[i][j]);
                        acc[i][j] = bv[j];
                    }
                epi_cnt = 0;                               // unknown number of stores: the next waits assume none (over-wait)
            }
            m_c = 0; ++m_tile; epi_age = 0;
            if (PROBE) { pc6 += cycle_now() - t2; pc7 += 1; }
        }
    }
    if (PROBE && probe != nullptr && tid == 0) {
        long long* o = probe + blockIdx.x * 10;
        o[0] = pc0; o[1] = pc1; o[2] = pc2; o[3] = pc3; o[4] = pc4; o[5] = pc5; o[6] = pc6; o[7] = pc7;
        o[8] = tr0;
        o[9] = realtime_now();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// gemm_nnl_kernel: gemm_nnq_kernel with a LOADER wave.  Workgroup = 5 waves: waves 0-3 are the four 48-column MFMA groups
// of gemm_nnq_kernel, wave 4 issues all 20 LDS-DMAs of a chunk (8 activation + 12 weight pieces) and owns the vmcnt
// bookkeeping.  An LDS-DMA costs its issuing wave ~60-180 cycles (MI355X_MICROARCH.md; ablation in tools/micro/gemm_lab.hip:
// 5 DMAs per MFMA wave and chunk cost 7-14 % of the kernel even with cache-hot sources), which is time the wave cannot
// issue MFMAs in; the loader wave has nothing else to do.  The MFMA waves then need no DMA state and fit 168 registers
// (3 waves per SIMD: 2 x 4 MFMA waves + 2 loaders per CU).  Same operands, packs, K order, tiles and results as
// gemm_nnq_kernel.  Barriers: one in front of the first chunk, one per chunk after it -- in both roles.
template <int NS>
__global__ __launch_bounds__(320, 3) void gemm_nnl_kernel(SegPtrs segs, int nseg, int F, int R,
                                                         const float* __restrict__ Bq, int nct_total,
                                                         const float* __restrict__ bias, float* __restrict__ C, int ldc, int O,
                                                         int btT, int btB, int btN) {
    constexpr int NB = 12, AF = 128 * 16, ST = kNnqStageFloats;
    static_assert(NS >= 3 && NS <= 5, "ring depth");
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
    const NnqOrder ko = make_nnq_order(nseg, F);
    const int nch = ko.nch;
    const int RT = ceil_div(R, 16);
    const int rt0 = (int)((long long)blockIdx.x * RT / gridDim.x), rt1 = (int)((long long)(blockIdx.x + 1) * RT / gridDim.x);
    const int nrows = rt1 - rt0;                           // row tiles of this workgroup
    if (nrows <= 0) return;
    const int ntile = ceil_div(nrows, 8), nrt_last = nrows - 8 * (ntile - 1);
    const int ct0 = blockIdx.y * NB;
    const int Q = ntile * nch;

    if (w == 4) {
        // ---- loader wave --------------------------------------------------------------------------------------------
        const int lg = lane >> 4;
        const wbuf_t rb = make_wbuf(Bq + (size_t)ct0 * 256);
        const unsigned b_voff = (unsigned)lane * 16u;       // + 1 KB per column tile, + chunk offset: scalar
        const int nb = nct_total - ct0 < NB ? nct_total - ct0 : NB;   // column tiles of this block (the rest re-fetch the last one)
        const int a_piece = (lane & 3) ^ nnq_gsw(lg);      // logical 16-byte piece this lane fetches (rows 16j + lane/4)
        const float* tptr[2];                              // tail chunks: this lane's plane + column (see gemm_nnq_kernel)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) {
            const int tp = tc * 4 + a_piece;
            int seg = 0, f = 0;
            if (tp < nseg * ko.b) { seg = tp / ko.b; f = ko.a * 16 + (tp - seg * ko.b) * 4; }
            tptr[tc] = segs.p[seg] + f;
        }
        unsigned a_voff[8];
        auto tile_rows = [&](int tile) __attribute__((always_inline)) {
            const int row0 = (rt0 + 8 * tile) * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int r = row0 + 16 * j + (lane >> 2);
                if (r >= R) r = R - 1;
                if (btT > 0) {                             // batch-major segments (see gemm_nn_dma_kernel)
                    const int sm_ = r / btN, n = r - sm_ * btN, t = sm_ / btB, b = sm_ - t * btB;
                    r = (b * btT + t) * btN + n;
                }
                a_voff[j] = ((unsigned)r * F + 4 * a_piece) * 4u;
            }
        };
        int d_tile = 0, d_c = 0, d_seg = 0, d_kc = 0, d_stage = 0;
        tile_rows(0);
        auto issue_chunk = [&]() __attribute__((always_inline)) {
            float* base = sm + d_stage * ST;
            if (d_c < ko.nmain) {
                const wbuf_t ra = make_wbuf(segs.p[d_seg]);
#pragma unroll
                for (int j = 0; j < 8; ++j) wbuf_dma16(ra, base + j * 256, a_voff[j], (unsigned)d_kc * 4u);
                d_kc += 16;
                if (d_kc == ko.a * 16) { d_kc = 0; ++d_seg; }
            } else {
                const char* p = reinterpret_cast<const char*>(d_c == ko.nmain ? tptr[0] : tptr[1]) - 16 * a_piece;
#pragma unroll
                for (int j = 0; j < 8; ++j) lds_dma16(base + j * 256, reinterpret_cast<const float*>(p + a_voff[j]));
            }
            const unsigned bso = (unsigned)(d_c * nct_total) * 1024u;
#pragma unroll
            for (int j = 0; j < NB; ++j) wbuf_dma16(rb, base + AF + j * 256, b_voff, bso + 1024u * (unsigned)(j < nb ? j : nb - 1));
            d_stage = d_stage + 1 == NS ? 0 : d_stage + 1;
            if (++d_c == nch) {
                d_c = 0; d_seg = 0; d_kc = 0;
                if (++d_tile < ntile) tile_rows(d_tile);
            }
        };
#pragma unroll
        for (int p = 0; p < NS - 1; ++p)
            if (p < Q) issue_chunk();
        EEG_VM_WAIT_BARRIER(0);                            // chunk 0 (and the rest of the prologue) landed
        for (int q = 0; q + 1 < Q; ++q) {
            // chunk q+1 must have landed; chunks q+2 .. q+NS-2 (20 DMAs each) may stay in flight.  After the barrier every
            // MFMA wave has finished reading chunk q-1, whose stage takes chunk q+NS-1
            if (Q - 2 - q >= NS - 3) {
                if (NS == 3) EEG_VM_WAIT_BARRIER(0);
                if (NS == 4) EEG_VM_WAIT_BARRIER(20);
                if (NS == 5) EEG_VM_WAIT_BARRIER(40);
            } else {
                EEG_VM_WAIT_BARRIER(0);
            }
            if (q + NS - 1 < Q) issue_chunk();
        }
        return;
    }

    // ---- MFMA waves ---------------------------------------------------------------------------------------------------
    const int lr = lane & 15, lg = lane >> 4;
    const int c_col = 16 * (ct0 + 3 * w) + 4 * lg;         // first of this lane's 3 x 4 output columns (+ 16 j)
    const bool cols_full = 16 * (ct0 + 3 * w + 3) <= O;
    const wbuf_t rc = make_wbuf(C);
    const int a_lds = lr * 16 + 4 * (lg ^ nnq_gsw((lr >> 2) & 3));
    const int b_lds = AF + 3 * w * 256 + lane * 4;
    f32x4 acc[8][3], oa[8], ob[3], obn[3];
    // accumulators start from the bias: it is (re)loaded from global memory where a tile starts (3 cache-hot 16-byte loads
    // per lane, issued in front of the previous tile's stores) instead of living in 12 registers -- these waves have no
    // LDS-DMA in flight, so ordinary loads and their compiler-placed waits are harmless here
    auto load_bias = [&](f32x4 (&bv)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr && c_col + 16 * j + 3 < O) bv[j] = *reinterpret_cast<const f32x4*>(bias + c_col + 16 * j);
        }
    };
    {
        f32x4 bv[3];
        load_bias(bv);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = bv[j];
    }
    EEG_LDS_BARRIER();                                     // chunk 0 landed (the loader waited for it)
#pragma unroll
    for (int i = 0; i < 8; ++i) oa[i] = *reinterpret_cast<const f32x4*>(sm + a_lds + i * 256);
#pragma unroll
    for (int j = 0; j < 3; ++j) ob[j] = *reinterpret_cast<const f32x4*>(sm + b_lds + j * 256);
    int r_stage = 1, m_c = 0, m_tile = 0;
    for (int q = 0; q < Q; ++q) {
        if (q + 1 < Q) EEG_LDS_BARRIER();                  // chunk q+1 landed
        const float* st = sm + r_stage * ST;               // (after the last chunk: a stale stage, unused)
#pragma unroll
        for (int j = 0; j < 3; ++j) obn[j] = *reinterpret_cast<const f32x4*>(st + b_lds + j * 256);
        const int nrt = m_tile == ntile - 1 ? nrt_last : 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < nrt) {                                 // (a partial tile multiplies its own row tiles only)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(ob[j][s], oa[i][s], acc[i][j]);   // transposed issue
            }
            oa[i] = *reinterpret_cast<const f32x4*>(st + a_lds + i * 256);   // refilled in place from chunk q+1
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) ob[j] = obn[j];
        r_stage = r_stage + 1 == NS ? 0 : r_stage + 1;
        if (++m_c == nch) {                                // the tile of chunk q is complete
            const int row0 = (rt0 + 8 * m_tile) * 16;
            f32x4 bv[3];
            load_bias(bv);
            if (nrt == 8 && row0 + 128 <= R && cols_full) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        wbuf_st4(rc, (unsigned)(lr * ldc + c_col + 16 * j), (unsigned)(row0 + 16 * i) * (unsigned)ldc, acc[i][j]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (i < nrt && row0 + 16 * i + lr < R && c_col + 16 * j < O)
                            wbuf_st4(rc, (unsigned)(lr * ldc + c_col + 16 * j), (unsigned)(row0 + 16 * i) * (unsigned)ldc, acc[i][j]);
            }
            EEG_SCHED_FENCE();                             // (a store's data registers must not be rewritten right behind it)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = bv[j];
            m_c = 0; ++m_tile;
        }
    }
}

}  // namespace eeg
