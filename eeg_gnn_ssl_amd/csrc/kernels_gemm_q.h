// Round-3 hoisted NN GEMM (gemm_nnq_kernel): C[R x O] = [A_0 | A_1 | ...] * W + bias on v_mfma_f32_16x16x4_f32.
//
// What changed against gemm_nn_dma_kernel (kernels_gemm.h), and why (profiles/r03_*):
//  * PERSISTENT row ranges.  A workgroup owns a contiguous range of 16-row tiles (balanced to one row tile over the grid,
//    2 workgroups per CU) and walks it in 128-row tiles.  The chunk stream runs on across tile boundaries: the operands of
//    the next tile are in flight while the current tile is finished, and its C stores drain under the next tile's MFMAs
//    (the one-tile-per-workgroup kernel had every CU in its load prologue / store epilogue at the same time).
//  * LDS ring of NS stages filled by buffer_load_dwordx4 ... lds (descriptor + per-lane offset fixed per tile + scalar
//    offset per chunk: NO vector address arithmetic in the loop) and COUNTED s_waitcnt vmcnt(N): a chunk is waited for
//    NS-1 chunks after it was requested, and never with vmcnt(0) in the steady state.
//  * ds_read_b128 fragments.  The weight pack is quad-ordered (pack[((c*NCT + ct)*64 + lane)*4 + s] = W[k(c, lane>>4, s)][16ct
//    + (lane&15)]): one 16-byte read per lane feeds the four MFMAs of a 16-deep K chunk; the activations are read as A[row][4g
//    .. 4g+3] by lane group g, so MFMA s of a chunk contracts k = 16c + 4g + s on lane group g (a sum does not care about the
//    order).  11 LDS reads per 96 MFMAs instead of 40, and they are requested one chunk ahead of the MFMAs that use them.
//  * 4 waves = 4 column groups of 48 (3 column tiles) x all 8 row tiles of the 128-row tile: a partial last tile costs
//    its row tiles only, which is what makes the balanced row ranges worth having.
//  * K order: the 16-deep chunks never straddle a hop plane: first the a = F/16 whole chunks of every plane, then the
//    leftover 16-byte pieces of all planes gathered into tail chunks (F = 100: 18 chunks + 1 tail chunk with one zero piece).
// The A tile in LDS is [128 rows][4 pieces of 16 B], piece p of row r stored at p ^ gsw((r >> 2) & 3) with gsw = {0,3,2,1}:
// the four 16-lane service groups of a ds_read_b128 ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS) then each cover the
// 64 banks exactly once.  The swizzle is applied on the SOURCE side of the DMA (which piece a lane fetches) and on the read.
// Reference semantics: model/cell.py:98-117 (the dense contraction of the diffusion convolution, x-part).
#pragma once
#include "kernels_gemm.h"

namespace eeg {

struct NnqOrder { int nseg, F, a, b, nmain, ntail, nch; };
__host__ __device__ inline NnqOrder make_nnq_order(int nseg, int F) {
    NnqOrder o;
    o.nseg = nseg; o.F = F; o.a = F / 16; o.b = (F / 4) % 4;
    o.nmain = nseg * o.a; o.ntail = (nseg * o.b + 3) / 4; o.nch = o.nmain + o.ntail;
    return o;
}
// logical K index (seg*F + f) of element s of 16-byte piece p of chunk c; -1 = zero padding
__host__ __device__ inline int nnq_k_of(const NnqOrder& o, int c, int p, int s) {
    if (c < o.nmain) return (c / o.a) * o.F + (c % o.a) * 16 + 4 * p + s;
    const int tp = (c - o.nmain) * 4 + p;
    if (tp >= o.nseg * o.b) return -1;
    return (tp / o.b) * o.F + o.a * 16 + (tp % o.b) * 4 + s;
}
constexpr int kNnqStageFloats = 128 * 16 + 12 * 256;   // A tile + 12 column tiles of the quad pack = 20 KB
__host__ __device__ constexpr int nnq_gsw(int x) { return (4 - x) & 3; }

#if !defined(EEG_SIMT_EMU)
__device__ __forceinline__ void wbuf_dma16(wbuf_t b, float* lds_wave_base, unsigned voff_bytes, unsigned soff_bytes) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff_bytes, soff_bytes, 0, 0);
}
#define EEG_VM_WAIT_BARRIER(n) asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory")
#else
__device__ __forceinline__ void wbuf_dma16(wbuf_t b, float* lds_wave_base, unsigned voff_bytes, unsigned soff_bytes) {
    memcpy(lds_wave_base + 4 * (threadIdx.x & 63), reinterpret_cast<const char*>(b.p) + voff_bytes + soff_bytes, 16);
}
#define EEG_VM_WAIT_BARRIER(n) __syncthreads()
#endif

template <int V> struct IntC { static constexpr int value = V; };

// ABL (lab only): 1 = no C stores, 2 = no DMA after the prologue, 3 = both.
// Requires: O % 4 == 0, ldc % 4 == 0, F % 4 == 0, at most 2 tail chunks (make_nnq_order(nseg, F).ntail <= 2), every
// segment and C smaller than 4 GB (32-bit buffer offsets).  LDS: NS stages of 20 KB.
template <int NS, int ABL>
__global__ __launch_bounds__(256, 2) void gemm_nnq_kernel(SegPtrs segs, int nseg, int F, int R,
                                                         const float* __restrict__ Bq, int nct_total,
                                                         const float* __restrict__ bias, float* __restrict__ C, int ldc, int O,
                                                         int btT, int btB, int btN) {
    constexpr int NB = 12, AF = 128 * 16, ST = kNnqStageFloats, NST = 24;
    static_assert(NS >= 2 && NS <= 5, "ring depth");
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), lr = lane & 15, lg = lane >> 4;
    const NnqOrder ko = make_nnq_order(nseg, F);
    const int nch = ko.nch;
    const int RT = ceil_div(R, 16);
    const int rt0 = (int)((long long)blockIdx.x * RT / gridDim.x), rt1 = (int)((long long)(blockIdx.x + 1) * RT / gridDim.x);
    const int ntile = ceil_div(rt1 - rt0, 8);
    if (ntile <= 0) return;
    const int nrt_last = (rt1 - rt0) - 8 * (ntile - 1);
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
        const int row0 = (rt0 + 8 * tile) * 16;
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
    auto issue_dma = [&]() __attribute__((always_inline)) {
        float* base = sm + d_stage * ST;
        if (d_c < ko.nmain) {
            const wbuf_t ra = make_wbuf(segs.p[d_seg]);
            wbuf_dma16(ra, base + w * 256, a_voff[0], (unsigned)d_kc * 4u);
            wbuf_dma16(ra, base + (w + 4) * 256, a_voff[1], (unsigned)d_kc * 4u);
            d_kc += 16;
            if (d_kc == ko.a * 16) { d_kc = 0; ++d_seg; }
        } else {
            const char* p = reinterpret_cast<const char*>(d_c == ko.nmain ? tptr[0] : tptr[1]) - 16 * a_piece;
            lds_dma16(base + w * 256, reinterpret_cast<const float*>(p + a_voff[0]));
            lds_dma16(base + (w + 4) * 256, reinterpret_cast<const float*>(p + a_voff[1]));
        }
        const unsigned bso = (unsigned)(d_c * nct_total) * 1024u;
        wbuf_dma16(rb, base + AF + w * 256, b_voff[0], bso);
        wbuf_dma16(rb, base + AF + (w + 4) * 256, b_voff[1], bso);
        wbuf_dma16(rb, base + AF + (w + 8) * 256, b_voff[2], bso);
        d_stage = d_stage + 1 == NS ? 0 : d_stage + 1;
        if (++d_c == nch) {
            d_c = 0; d_seg = 0; d_kc = 0;
            if (++d_tile < ntile) tile_rows(d_tile);
        }
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
            if ((ABL & 2) == 0 && q + NS - 1 < Q) issue_dma();
        }
        // (the last chunk of the range re-reads a stale stage into registers nobody uses: no branch around the reads)
        const float* st = sm + r_stage * ST;
#pragma unroll
        for (int j = 0; j < 3; ++j) obn[j] = *reinterpret_cast<const f32x4*>(st + b_lds + j * 256);
        const int nrt = m_tile == ntile - 1 ? nrt_last : 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < nrt) {                                 // (a partial last tile multiplies its own row tiles only)
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
        ++epi_age;
        if (++m_c == nch) {                                // the tile of chunk q is complete
            const int row0 = (rt0 + 8 * m_tile) * 16;
            if (ABL & 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
#if !defined(EEG_SIMT_EMU)
                        asm volatile("" ::"v"(acc[i][j]));
#endif
                    }
                epi_cnt = 0;
            } else if (nrt == 8 && row0 + 128 <= R && cols_full) {   // exactly NST unconditional stores: the counted waits rely on it
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        wbuf_st4(rc, (unsigned)(lr * ldc + c_col + 16 * j), (unsigned)(row0 + 16 * i) * (unsigned)ldc, acc[i][j]);
                epi_cnt = NST;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (i < nrt && row0 + 16 * i + lr < R && c_col + 16 * j < O)
                            wbuf_st4(rc, (unsigned)(lr * ldc + c_col + 16 * j), (unsigned)(row0 + 16 * i) * (unsigned)ldc, acc[i][j]);
                epi_cnt = 0;                               // unknown number of stores: the next waits assume none (over-wait)
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = bv[j];
            m_c = 0; ++m_tile; epi_age = 0;
        }
    }
}

}  // namespace eeg
