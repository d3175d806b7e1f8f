// Round-3 hoisted NN GEMM (gemm_nnr_kernel; its lab predecessor gemm_nnq_kernel and the loader-wave variant gemm_nnl_kernel live
// with the lab, tools/micro/gemm_lab_kernels.h): C[R x O] = [A_0 | A_1 | ...] * W + bias on v_mfma_f32_16x16x4_f32.
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
#include "nnq_order.h"

namespace eeg {

__host__ __device__ constexpr int nnq_gsw(int x) { return (4 - x) & 3; }


template <int V> struct IntC { static constexpr int value = V; };
// a value the optimiser cannot see through: keeps rare-path computations from being hoisted out of a hot loop (and
// parked in registers the loop needs)
__device__ __forceinline__ int opaque(int v) {
    EEG_PIN(v);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// gemm_nnr_kernel (the shipped NN kernel): the design above with the WEIGHTS going global -> registers (the first version,
// gemm_nnq_kernel, staged them in LDS beside the activations).  In the 4 x 48-column
// wave layout a wave is the only reader of its 3 weight column tiles, so staging them in LDS buys nothing: here every lane
// fetches its three 16-byte fragments of the next chunk straight from the quad pack (L2-resident, one coalesced 1-KB load per
// tile) while the current chunk is multiplied.  Per wave and chunk: 2 LDS-DMAs (activations) + 3 plain loads instead of 5
// LDS-DMAs + 3 ds_reads, an 8-KB stage instead of 20 KB (ring of 4 = 32 KB + the bias).  Lab: 0.728 -> 0.751 (K = 192),
// 0.743 -> 0.766 (K = 300) of the fp32-MFMA peak.
// Queue discipline: per iteration a wave issues, in this order, the 3 weight loads of chunk q+1 and the 2 activation DMAs of
// chunk q+3; the weight fragments are ordinary (compiler-visible) buffer loads, so the compiler itself waits for them where
// they are first used -- the end of the iteration -- and, the vector-memory queue being in order (tools/micro/order_lab.hip),
// everything older has arrived with them: the DMAs of chunk q+2 and the stores of the previous tile.  (Beside LDS-DMAs
// hipcc makes that wait a vmcnt(0), i.e. the DMAs of chunk q+3 are drained too; they have had the whole iteration.)  The
// explicit s_waitcnt in front of the copy restates what the next barrier relies on, whatever the compiler chooses.
// A first version issued the weight loads as inline asm with hand-counted waits: sporadically wrong tiles (the register
// allocator is free to place asm outputs where the asynchronous return collides with its own copies) -- do not do that.
template <int NS, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_nnr_kernel(SegPtrs segs, int nseg, int F, int R,
                                                         const float* __restrict__ Bq, int nct_total,
                                                         const float* __restrict__ bias, float* __restrict__ C, int ldc, int O,
                                                         int btT, int btB, int btN, int skew) {
    constexpr int NB = 12, ST = 128 * 16;                  // a stage = the 128 x 16 activation tile
    static_assert(NS == 4, "the queue discipline above assumes a ring of 4");
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), lr = lane & 15, lg = lane >> 4;
    const NnqOrder ko = make_nnq_order(nseg, F);
    const int nch = ko.nch;
    const int RT = ceil_div(R, 16);
    // row tiles of this workgroup.  skew (per mille): the two workgroups of a CU are ids b and b + G/2; the older one (b) wins
    // every arbitration for the matrix pipe and would finish ~5 % ahead of the other, which then runs alone: the lower half
    // of the grid gets (1000 + skew) / 1000 of the average share, the upper half the rest
    const int G = gridDim.x, half = G / 2, bid = blockIdx.x;
    int rt0, rt1;
    if (skew == 0 || (G & 1) != 0) {
        rt0 = (int)((long long)bid * RT / G); rt1 = (int)((long long)(bid + 1) * RT / G);
    } else {
        const int lo = (int)((long long)RT * (1000 + skew) / 2000);      // row tiles of the lower half of the grid
        if (bid < half) { rt0 = (int)((long long)bid * lo / half); rt1 = (int)((long long)(bid + 1) * lo / half); }
        else { rt0 = lo + (int)((long long)(bid - half) * (RT - lo) / half); rt1 = lo + (int)((long long)(bid - half + 1) * (RT - lo) / half); }
    }
    const int nrows = rt1 - rt0;
    if (nrows <= 0) return;
    const int ntile = ceil_div(nrows, 8), nrt_last = nrows - 8 * (ntile - 1);
    const int ct0 = blockIdx.y * NB;
    const int Q = ntile * nch;
    float* const bias_s = sm + NS * ST;                    // 192 floats behind the ring
    if (tid < 192) bias_s[tid] = (bias != nullptr && 16 * ct0 + tid < O) ? bias[16 * ct0 + tid] : 0.f;

    // ---- sources ----------------------------------------------------------------------------------------------------
    const int ctw = ct0 + 3 * w + 2 < nct_total ? ct0 + 3 * w : (nct_total >= 3 ? nct_total - 3 : 0);   // (a partial block re-reads valid tiles)
    const wbuf_t rb = make_wbuf(Bq + (size_t)ctw * 256);
    const unsigned b_voff = (unsigned)lane * 16u;
    const int a_piece = (lane & 3) ^ nnq_gsw(lg);
    const float* tptr[2];
#pragma unroll
    for (int tc = 0; tc < 2; ++tc) {
        const int tp = tc * 4 + a_piece;
        int seg = 0, f = 0;
        if (tp < nseg * ko.b) { seg = tp / ko.b; f = ko.a * 16 + (tp - seg * ko.b) * 4; }
        tptr[tc] = segs.p[seg] + f;
    }
    int d_tile = 0, d_c = 0, d_seg = 0, d_kc = 0, d_stage = 0;
    unsigned a_voff[2];
    auto tile_rows = [&](int tile) __attribute__((always_inline)) {
        const int row0 = (rt0 + 8 * tile) * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r = row0 + 16 * (w + 4 * i) + (lane >> 2);
            if (r >= R) r = R - 1;
            if (btT > 0) {
                const int sm_ = r / btN, n = r - sm_ * btN, t = sm_ / btB, b = sm_ - t * btB;
                r = (b * btT + t) * btN + n;
            }
            a_voff[i] = ((unsigned)r * F + 4 * a_piece) * 4u;
        }
    };
    tile_rows(0);
    auto issue_a = [&]() __attribute__((always_inline)) {  // the 2 activation DMAs of this wave for the chunk at the DMA cursor
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
        d_stage = d_stage + 1 == NS ? 0 : d_stage + 1;
        if (++d_c == nch) {
            d_c = 0; d_seg = 0; d_kc = 0;
            if (++d_tile < ntile) tile_rows(d_tile);
        }
    };

    // ---- compute side -----------------------------------------------------------------------------------------------
    const int c_col = 16 * (ct0 + 3 * w) + 4 * lg;
    const bool cols_full = 16 * (ct0 + 3 * w + 3) <= O;
    const wbuf_t rc = make_wbuf(C);
    const int a_lds = lr * 16 + 4 * (lg ^ nnq_gsw((lr >> 2) & 3));
    f32x4 acc[8][3], oa[8], ob[3], obn[3];
    // prologue: weights of chunk 0 and 1, activations of chunks 0 .. 2
#pragma unroll
    for (int j = 0; j < 3; ++j) ob[j] = wbuf_ld4(rb, b_voff / 4 + 256 * j, 0u);
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < Q) issue_a();
    __syncthreads();                                       // bias_s + the prologue DMAs of all waves
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(bias_s + 16 * (3 * w + j) + 4 * lg);
#pragma unroll
    for (int i = 0; i < 8; ++i) oa[i] = *reinterpret_cast<const f32x4*>(sm + a_lds + i * 256);
    int r_stage = 1, m_c = 0, m_tile = 0, b_c = 1;         // b_c: chunk index (within its tile) of the weights requested next
    if (b_c == nch) b_c = 0;
    for (int q = 0; q < Q; ++q) {
        const bool more = q + 1 < Q, dma = q + NS - 1 < Q;
        if (more) EEG_LDS_BARRIER();                       // chunk q+1 landed in every wave (each waited at the end of its iteration q-1)
        if (more) {
            const unsigned bso = (unsigned)(b_c * nct_total) * 1024u;
#pragma unroll
            for (int j = 0; j < 3; ++j) obn[j] = wbuf_ld4(rb, b_voff / 4 + 256 * j, bso / 4);
            b_c = b_c + 1 == nch ? 0 : b_c + 1;
        }
        if (dma) issue_a();                                // chunk q+3 into the stage of chunk q-1 (every wave is past its reads)
        EEG_SCHED_FENCE();
        const float* st = sm + r_stage * ST;               // (after the last chunk: a stale stage, unused)
        const int nrt = m_tile == ntile - 1 ? nrt_last : 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < nrt) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(ob[j][s], oa[i][s], acc[i][j]);   // transposed issue
            }
            oa[i] = *reinterpret_cast<const f32x4*>(st + a_lds + i * 256);   // refilled in place from chunk q+1
        }
        EEG_SCHED_FENCE();
        // the weight fragments of chunk q+1 (and with them everything requested before: the DMAs of chunk q+2, older stores)
        if (more) {
            // (the compiler waits for the weight loads where they are first used -- here -- with the count of what it saw
            //  issued behind them, i.e. vmcnt(2): the explicit wait below only restates what the next barrier relies on)
            if (dma) EEG_VM_WAIT(2); else EEG_VM_WAIT(0);
#pragma unroll
            for (int j = 0; j < 3; ++j) ob[j] = obn[j];
        }
        r_stage = r_stage + 1 == NS ? 0 : r_stage + 1;
        if (++m_c == nch) {                                // the tile of chunk q is complete
            const int row0 = (rt0 + 8 * m_tile) * 16;
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
                for (int j = 0; j < 3; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(bias_s + 16 * (3 * w + j) + 4 * lg);
            m_c = 0; ++m_tile;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round-3 TN GEMM (weight gradients): partial[split][k][o] = sum over the rows of the split of A[r][k] * dY[r][ycol0 + o].
//
// Against gemm_tn_dma_kernel: (i) a workgroup owns a (2*KT*16) x (2*OT*16) output block (192 x 192 = the WHOLE gradient of a
// 64-unit cell at K = 192: A and dY are read once, nothing is re-read per k-block) for a long row range (2 workgroups per
// CU, persistent over their split); (ii) the transposed MFMA operands come from LDS as ds_read_b128 / b64: lane i of a
// 16-lane group reads 4 (2) CONSECUTIVE columns of one row and the 4 (2) MFMAs that use them own the column sets {4i + e}
// ({2i + e}) -- a permuted tile <-> column assignment that only the epilogue has to know; 4 LDS reads per 36 MFMAs instead
// of 12 per 12; (iii) dY streams through a buffer descriptor (no vector address arithmetic), and so does A where the hop
// planes are 64 wide (PLANAR: one plane per DMA instruction); otherwise A goes through per-lane running pointers (its
// 16-byte pieces come from different hop planes); (iv) LDS ring of 3 stages, the chunk barrier sits in front of the LAST
// k-step of a chunk so that the first fragments of the next chunk are read one k-step ahead of their MFMAs.
// LDS image of an operand chunk (RC rows, both wave slices wv = 0, 1 of T tiles each): three sub-arrays so that each read
// width is conflict-free with plain immediate offsets:  P128 [2][RC][64*n4]  (row stride = 0 mod 64 dwords: the four 16-lane
// service groups of a ds_read_b128 each cover all 64 banks),  P64 [RC][2][32] and P32 [RC][2][16] with the two wave slots
// of ODD rows swapped (lane groups kk and kk+1 of one half-wave read consecutive rows: the swap puts them on different banks).
// The images are filled by LDS-DMA (lane-linear), i.e. the layout is realised on the SOURCE side: lane l of DMA d fetches
// the global piece that belongs at image position 256 d + 4 l.
template <int T> struct TnqSlice {
    static constexpr int n4 = T / 4, rem = T % 4, has64 = rem >= 2 ? 1 : 0, has32 = rem & 1;
    static constexpr int w128 = 64 * n4, base64 = w128, base32 = base64 + 32 * has64;
    static constexpr int ngroups = n4 + has64 + has32;
    // slice-local column of element e of lane i's read in group g (g < n4: b128, then b64, then b32)
    __host__ __device__ static constexpr int col(int g, int i, int e) {
        return g < n4 ? 64 * g + 4 * i + e : (has64 && g == n4 ? base64 + 2 * i + e : base32 + i);
    }
    __host__ __device__ static constexpr int width(int g) { return g < n4 ? 4 : (has64 && g == n4 ? 2 : 1); }
    // column inside the (2 * T * 16)-wide block of slice-local column u of wave slice wv.  Natural order: wv * 16 T + u.
    // PLANAR (64-wide planes, T in {2, 4, 6}): the b128 group of slice 0 is the first plane of the block, that of slice 1 the
    // last one, and the two b64 groups are the halves of the middle plane -- every group lies inside ONE plane.
    template <bool PLANAR> __host__ __device__ static constexpr int block_col(int wv, int u) {
        if (!PLANAR) return wv * 16 * T + u;
        return u < w128 ? (wv == 0 ? 0 : 64 * (T / 2 - 1)) + u : 64 * n4 + 32 * wv + (u - base64);
    }
};
// (row, wave slice, slice-local column) of image position pos (floats, multiple of 4) of an operand with T tiles per slice
template <int T, int RC>
__device__ __forceinline__ void tnq_image_pos(int pos, int& row, int& wv, int& u) {
    using S = TnqSlice<T>;
    constexpr int n128 = RC * 2 * S::w128, n64 = RC * 64 * S::has64, w = S::w128 > 0 ? S::w128 : 1;
    if (pos < n128) {
        wv = pos / (RC * w);
        const int c = pos - wv * RC * w;
        row = c / w; u = c - row * w;
    } else if (pos < n128 + n64) {
        pos -= n128; row = pos >> 6;
        const int c = pos & 63;
        wv = (c >> 5) ^ (row & 1); u = S::base64 + (c & 31);
    } else {
        pos -= n128 + n64; row = pos >> 5;
        const int c = pos & 31;
        wv = (c >> 4) ^ (row & 1); u = S::base32 + (c & 15);
    }
}

// BT: the A segments are batch-major (btB clips x btT steps x btN nodes, see gemm_nn_dma_kernel); dY is always time-major.
// PLANAR: F == 64 and KT in {2, 4, 6} (a k-block = KT / 2 whole planes), not BT.  Requires Ov == 32 * OT (whole column block).
// TAIL = false: R % RC == 0 and rows_per_split % RC == 0 (no partial chunk anywhere): the clamp / zero-fill paths are compiled out.
// gemm_tnq_rows: the rows [rbeg, rend) of the operands (rend - rbeg a multiple of RC unless TAIL) -> partial slot `split`.
template <int KT, int OT, int RC, bool BT, bool PLANAR, bool TAIL>
__device__ __forceinline__ void gemm_tnq_rows(const SegPtrs& segs, int nseg, int F, int R,
                                              const float* __restrict__ dY, int ldy, int ycol0, int Ov,
                                              float* __restrict__ partial,
                                              int btT, int btB, int btN, int flags, const int kblock, const int split,
                                              const int rbeg, const int rend) {
    using SA = TnqSlice<KT>;
    using SY = TnqSlice<OT>;
    constexpr int NS = 3, KS = RC / 4;
    constexpr int A_FLOATS = RC * 32 * KT, Y_FLOATS = RC * 32 * OT, ST = A_FLOATS + Y_FLOATS;
    constexpr int A_INS = A_FLOATS / 256, Y_INS = Y_FLOATS / 256, NIA = (A_INS + 3) / 4, NIY = (Y_INS + 3) / 4;
    static_assert(KS % 2 == 0 && A_FLOATS % 256 == 0 && Y_FLOATS % 256 == 0, "chunk shape");
    static_assert(SA::n4 <= 1 && SY::n4 <= 2, "slice widths");
    static_assert(!PLANAR || (!BT && KT % 2 == 0 && SA::has32 == 0), "planar blocks are whole 64-wide planes");
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6), wk = w >> 1, wo = w & 1, li = lane & 15, kk = lane >> 4;
    // (Round 5 measured the XCD-aware placement of gemm_tn_dma_kernel here too -- the k-blocks of a row split given linear ids
    //  congruent mod 8 so that the second reader of the dY rows hits that XCD's L2: FETCH_SIZE of the two-k-block layer-0 instance
    //  stayed at 423 601 KB-units per launch and its time at 0.295 ms; with ~4 MB in flight per XCD the partner's lines are gone
    //  before it asks.  Not kept: profiles/r05_c_pmc_traffic_cfg2_{default,plain_order}.json.)
    const int K = nseg * F, k0 = kblock * (32 * KT);
    const int Q = rend > rbeg ? ceil_div(rend - rbeg, RC) : 0;

    // ---- DMA side ------------------------------------------------------------------------------------------------------
    // A DMA instruction covers 256 consecutive floats of a sub-array = 4 rows x 64 positions (P128, P64) or 8 rows x 32 (P32),
    // so the lane part of a source offset is the same for every DMA of a sub-array and the rest is wave-uniform:
    //   dY (and A when PLANAR): buffer descriptor + ONE per-lane offset per sub-array + a scalar offset per DMA;
    //   A otherwise: a running 64-bit pointer per DMA of this lane's piece (BT: pointer of storage row 0 + (clip, step, node)
    //   counters of the lane's current time-major row).
    constexpr int NA128 = RC * 2 * SA::w128 / 256, NA64 = RC * 64 * SA::has64 / 256;
    constexpr int NY128 = RC * 2 * SY::w128 / 256, NY64 = RC * 64 * SY::has64 / 256;
    // lane parts (floats): row-in-DMA * ld + column; P64 / P32 pick the wave slot by the row parity (see the image layout).
    // They are a handful of integer operations on the lane id and are RE-computed at every use (opaque(): not hoisted) --
    // as loop-invariant registers they were what the 192 x 192 instance spilled.
    auto lane_part = [&](int kind, int ld, int slot_w, int b64, int b32) __attribute__((always_inline)) -> unsigned {
        const int ln = opaque(lane);
        const int l16 = ln >> 4, c16 = 4 * (ln & 15), l8 = ln >> 3, c8 = 4 * (ln & 7);
        if (kind == 0) return (unsigned)(l16 * ld + c16);
        if (kind == 1) return (unsigned)(l16 * ld + (((c16 >> 5) ^ (l16 & 1)) * slot_w + b64 + (c16 & 31)));
        return (unsigned)(l8 * ld + (((c8 >> 4) ^ (l8 & 1)) * slot_w + b32 + (c8 & 15)));
    };
    const wbuf_t ry = make_wbuf(dY + ycol0);
    const char* a_ptr[NIA];
    int mb[NIA], mt[NIA], mn[NIA];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        mb[i] = mt[i] = mn[i] = 0;
        a_ptr[i] = nullptr;
        if (PLANAR) continue;
        const int d = w + 4 * i < A_INS ? w + 4 * i : A_INS - 1;   // (a wave without a DMA of its own repeats the last one: same bytes)
        int row, wv, u;
        tnq_image_pos<KT, RC>(256 * d + 4 * lane, row, wv, u);
        int k = k0 + SA::template block_col<false>(wv, u);
        if (k >= K) k = K - 4;                             // columns past K: fetched, multiplied, never stored
        const int seg = k / F, f = k - seg * F;
        const float* base = segs.p[0];
#pragma unroll
        for (int m = 1; m < kMaxM; ++m)
            if (m < nseg && seg == m) base = segs.p[m];
        int r = rbeg + row;                                // (rows past R are clamped when they are requested)
        if (BT) {
            if (r >= R) r = R - 1;
            const int sm_ = r / btN;
            mn[i] = r - sm_ * btN; mt[i] = sm_ / btB; mb[i] = sm_ - mt[i] * btB;
            r = 0;
        }
        a_ptr[i] = reinterpret_cast<const char*>(base + f) + (size_t)r * F * 4;
    }
    int d_q = 0, d_stage = 0;
    // one DMA of a descriptor-addressed operand: sub-array / row block / wave slot from the DMA index (wave-uniform)
    auto dma_desc = [&](wbuf_t rs, float* dst, int d, int n128, int n64, int ld, int slot_w, int b64, int b32,
                        int r0, int slot_stride, bool tail) __attribute__((always_inline)) {
        unsigned rows_first, slot = 0;
        int kind;
        if (d < n128) {
            const int per = n128 / 2;
            slot = d / per; rows_first = 4 * (d - slot * per); kind = 0;
        } else if (d < n128 + n64) {
            rows_first = 4 * (d - n128); kind = 1;
        } else {
            rows_first = 8 * (d - n128 - n64); kind = 2;
        }
        unsigned vo = lane_part(kind, ld, slot_w, b64, b32) * 4u;
        const unsigned r1 = (unsigned)r0 + rows_first;
        if (tail) {                                        // rows past R re-read row R-1 (they are zeroed in LDS / never stored)
            const int rin = kind == 2 ? opaque(lane) >> 3 : opaque(lane) >> 4;
            const int over = (int)r1 + rin - (R - 1);
            if (over > 0) vo -= (unsigned)over * (unsigned)ld * 4u;
        }
        wbuf_dma16(rs, dst, vo, (r1 * (unsigned)ld + slot * (unsigned)slot_stride) * 4u);
    };
    auto issue_dma = [&]() __attribute__((always_inline)) {
        float* base = sm + d_stage * ST;
        const int r0 = rbeg + d_q * RC;
        const bool tail = TAIL && r0 + RC > R;             // only the last chunk of the last split
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int d = w + 4 * i < A_INS ? w + 4 * i : A_INS - 1;
            if (PLANAR) {
                // plane of DMA d: P128 slot 0 -> first plane of the block, slot 1 -> the last one, P64 -> the middle one
                int pl = d < NA128 ? (d < NA128 / 2 ? 0 : KT / 2 - 1) : SA::n4;
                pl += k0 / 64;
                if (pl >= nseg) pl = nseg - 1;             // planes past K: fetched, multiplied, never stored
                dma_desc(make_wbuf(segs.p[pl]), base + d * 256, d, NA128, NA64, F, 32, 0, 0, r0, 0, tail);
                continue;
            }
            const char* src;
            if (BT) {
                int mapped = (mb[i] * btT + mt[i]) * btN + mn[i];
                if (tail) {
                    int row, wv, u;
                    tnq_image_pos<KT, RC>(256 * d + 4 * opaque(lane), row, wv, u);
                    if (r0 + row >= R) mapped = R - 1;     // (the last time-major row is the last storage row)
                }
                src = a_ptr[i] + (size_t)mapped * F * 4;
                mn[i] += RC;
                while (mn[i] >= btN) {
                    mn[i] -= btN;
                    if (++mb[i] == btB) { mb[i] = 0; ++mt[i]; }
                }
            } else {
                src = a_ptr[i];
                if (tail) {
                    int row, wv, u;
                    tnq_image_pos<KT, RC>(256 * d + 4 * opaque(lane), row, wv, u);
                    if (r0 + row >= R) src -= (size_t)(r0 + row - (R - 1)) * F * 4;
                }
                a_ptr[i] += (size_t)RC * F * 4;
            }
            lds_dma16(base + d * 256, reinterpret_cast<const float*>(src));
        }
#pragma unroll
        for (int i = 0; i < NIY; ++i) {
            const int d = w + 4 * i < Y_INS ? w + 4 * i : Y_INS - 1;
            dma_desc(ry, base + A_FLOATS + d * 256, d, NY128, NY64, ldy, 16 * OT, SY::base64, SY::base32, r0, 16 * OT, tail);
        }
        d_stage = d_stage + 1 == NS ? 0 : d_stage + 1;
        ++d_q;
    };

    // ---- compute side -----------------------------------------------------------------------------------------------
    f32x4 acc[KT][OT];
    // fragment read offsets (floats) of k-step 0; k-step ks adds 4 * ks * (row stride of the sub-array)
    constexpr int A128 = 0, A64 = A128 + RC * 2 * SA::w128, A32 = A64 + RC * 64 * SA::has64;
    constexpr int Y128 = A_FLOATS, Y64 = Y128 + RC * 2 * SY::w128, Y32 = Y64 + RC * 64 * SY::has64;
    const int a128 = A128 + (wk * RC + kk) * SA::w128 + 4 * li;
    const int a64 = A64 + kk * 64 + ((wk ^ (kk & 1)) << 5) + 2 * li;
    const int a32 = A32 + kk * 32 + ((wk ^ (kk & 1)) << 4) + li;
    const int y128 = Y128 + (wo * RC + kk) * SY::w128 + 4 * li;
    const int y64 = Y64 + kk * 64 + ((wo ^ (kk & 1)) << 5) + 2 * li;
    const int y32 = Y32 + kk * 32 + ((wo ^ (kk & 1)) << 4) + li;
    float fa[2][KT], fy[2][OT];
    auto read_frags = [&](auto PAR, const float* st, int ks) __attribute__((always_inline)) {
        constexpr int p = decltype(PAR)::value;
#pragma unroll
        for (int g = 0; g < SA::n4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(st + a128 + 64 * g + ks * 4 * SA::w128);
#pragma unroll
            for (int e = 0; e < 4; ++e) fa[p][4 * g + e] = v[e];
        }
        if (SA::has64) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(st + a64 + ks * 4 * 64);
            fa[p][4 * SA::n4] = v[0]; fa[p][4 * SA::n4 + 1] = v[1];
        }
        if (SA::has32) fa[p][KT - 1] = st[a32 + ks * 4 * 32];
#pragma unroll
        for (int g = 0; g < SY::n4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(st + y128 + 64 * g + ks * 4 * SY::w128);
#pragma unroll
            for (int e = 0; e < 4; ++e) fy[p][4 * g + e] = v[e];
        }
        if (SY::has64) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(st + y64 + ks * 4 * 64);
            fy[p][4 * SY::n4] = v[0]; fy[p][4 * SY::n4 + 1] = v[1];
        }
        if (SY::has32) fy[p][OT - 1] = st[y32 + ks * 4 * 32];
    };
    auto mfma_step = [&](auto PAR) __attribute__((always_inline)) {
        constexpr int p = decltype(PAR)::value;
#pragma unroll
        for (int a = 0; a < KT; ++a)
#pragma unroll
            for (int b = 0; b < OT; ++b) acc[a][b] = mfma16(fa[p][a], fy[p][b], acc[a][b]);
    };
    auto zero_tail = [&](float* st, int valid) __attribute__((always_inline)) {   // dY rows >= valid of a chunk contribute nothing
        for (int e = opaque(tid); e < 2 * (RC - valid) * SY::w128; e += 256) {
            const int wv = e / ((RC - valid) * (SY::w128 > 0 ? SY::w128 : 1)), c = e - wv * (RC - valid) * SY::w128;
            st[Y128 + (wv * RC + valid) * SY::w128 + c] = 0.f;
        }
        for (int e = opaque(tid); e < (RC - valid) * 64 * SY::has64; e += 256) st[Y64 + valid * 64 + e] = 0.f;
        for (int e = opaque(tid); e < (RC - valid) * 32 * SY::has32; e += 256) st[Y32 + valid * 32 + e] = 0.f;
    };

    if (Q <= 0) {
#pragma unroll
        for (int a = 0; a < KT; ++a)
#pragma unroll
            for (int b = 0; b < OT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
        issue_dma();
        if (Q > 1) issue_dma();
        __syncthreads();                                   // both chunks landed (once per workgroup)
        if (TAIL && rbeg + RC > rend) { zero_tail(sm, rend - rbeg); __syncthreads(); }
        EEG_SCHED_FENCE();                                 // (the accumulators are born here, after the set-up arithmetic)
#pragma unroll
        for (int a = 0; a < KT; ++a)
#pragma unroll
            for (int b = 0; b < OT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        read_frags(IntC<0>(), sm, 0);
        int r_stage = 0;
        for (int q = 0; q < Q; ++q) {
            const float* st = sm + r_stage * ST;
#pragma unroll
            for (int ks = 0; ks + 1 < KS; ++ks) {
                if (ks & 1) { read_frags(IntC<0>(), st, ks + 1); EEG_SCHED_FENCE(); mfma_step(IntC<1>()); }
                else        { read_frags(IntC<1>(), st, ks + 1); EEG_SCHED_FENCE(); mfma_step(IntC<0>()); }
                EEG_SCHED_FENCE();
            }
            r_stage = r_stage + 1 == NS ? 0 : r_stage + 1;
            float* nx = sm + r_stage * ST;
            if (q + 1 < Q) {
                // chunk q+1 (requested a chunk ago) must have landed; after the barrier every wave is past its reads of
                // chunk q-1, whose stage takes chunk q+2
                EEG_VM_WAIT_BARRIER(0);
                if (q + 2 < Q && (flags & 4) == 0) issue_dma();          // (flags 4, lab: no DMA after the prologue)
                const int r1 = rbeg + (q + 1) * RC;
                if (TAIL && r1 + RC > rend) { zero_tail(nx, rend - r1); EEG_LDS_BARRIER(); }
            }
            read_frags(IntC<0>(), nx, 0);                  // (after the last chunk: a stale stage, unused)
            EEG_SCHED_FENCE();
            mfma_step(IntC<1>());                          // k-step KS-1 (KS is even)
            EEG_SCHED_FENCE();
        }
    }
    // ---- epilogue: partial[split][k][o], k / o through the tile <-> column maps of the two slices ----------------------------
    // k = k0 + block_col(wk, col(ga, 4 kk + r, ea)): the lane part (kk) goes into the per-lane offset, the (r, ea) part is a
    // wave-uniform row offset of the buffer store (requires K * Ov * 4 < 4 GB per split: it is a weight gradient)
    if (flags & 1) {                                       // lab: no partial stores (the accumulators stay live)
#pragma unroll
        for (int a = 0; a < KT; ++a)
#pragma unroll
            for (int b = 0; b < OT; ++b) EEG_USE(acc[a][b]);
        return;
    }
    const wbuf_t ro = make_wbuf(partial + (size_t)((flags & 2) ? split & 7 : split) * K * Ov);   // flags 2 (lab): 8 cache-resident slots
#pragma unroll
    for (int ga = 0; ga < SA::ngroups; ++ga) {
        const int kl = k0 + SA::template block_col<PLANAR>(wk, SA::col(ga, 4 * kk, 0));       // r = 0, ea = 0
        const int kstep_r = SA::col(ga, 1, 0) - SA::col(ga, 0, 0);                               // columns per lane index step
#pragma unroll
        for (int ea = 0; ea < SA::width(ga); ++ea) {
            const int a = ga < SA::n4 ? 4 * ga + ea : (SA::has64 && ga == SA::n4 ? 4 * SA::n4 + ea : KT - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int dk = r * kstep_r + ea;                                               // uniform part of k
                const bool kok = kl + dk < K;
#pragma unroll
                for (int gb = 0; gb < SY::ngroups; ++gb) {
                    const int o = wo * 16 * OT + SY::col(gb, li, 0);
                    const unsigned vo = (unsigned)(kl * Ov + o), so = (unsigned)(dk * Ov);
                    if (gb < SY::n4) {
                        if (kok && o + 3 < Ov) wbuf_st4(ro, vo, so, (f32x4){acc[a][4 * gb][r], acc[a][4 * gb + 1][r], acc[a][4 * gb + 2][r], acc[a][4 * gb + 3][r]});
                    } else if (SY::has64 && gb == SY::n4) {
                        if (kok && o + 1 < Ov) wbuf_st2(ro, vo, so, acc[a][4 * SY::n4][r], acc[a][4 * SY::n4 + 1][r]);
                    } else if (kok && o < Ov) {
                        wbuf_st1(ro, vo, so, acc[a][OT - 1][r]);
                    }
                }
            }
        }
    }
}

// row split `split` of rows_per_split rows (the last one ends at R)
template <int KT, int OT, int RC, bool BT, bool PLANAR, bool TAIL>
__device__ __forceinline__ void gemm_tnq_body(const SegPtrs& segs, int nseg, int F, int R,
                                              const float* __restrict__ dY, int ldy, int ycol0, int Ov,
                                              float* __restrict__ partial, int rows_per_split,
                                              int btT, int btB, int btN, int flags, const int kblock, const int split) {
    const int rbeg = split * rows_per_split;
    const int rend = (rbeg + rows_per_split < R) ? rbeg + rows_per_split : R;
    gemm_tnq_rows<KT, OT, RC, BT, PLANAR, TAIL>(segs, nseg, F, R, dY, ldy, ycol0, Ov, partial, btT, btB, btN, flags, kblock, split, rbeg, rend);
}

template <int KT, int OT, int RC, bool BT, bool PLANAR, bool TAIL>
__global__ __launch_bounds__(256, 2) void gemm_tnq_kernel(SegPtrs segs, int nseg, int F, int R,
                                                         const float* __restrict__ dY, int ldy, int ycol0, int Ov,
                                                         float* __restrict__ partial, int rows_per_split,
                                                         int btT, int btB, int btN, int flags) {
    gemm_tnq_body<KT, OT, RC, BT, PLANAR, TAIL>(segs, nseg, F, R, dY, ldy, ycol0, Ov, partial, rows_per_split, btT, btB, btN, flags,
                                                (int)blockIdx.x, (int)blockIdx.y);
}

// Round 5: the two h-part weight-gradient GEMMs of a cell -- hops(h)^T [dR|dU] (128 columns, MFMA-bound) and
// hops(r*h)^T dC (64 columns, waits for its operands) -- as ONE launch of 2 * nsplit workgroups per k-block: the two problems
// share K, the row splits and the dY rows; the jobs alternate over the two workgroup slots of a CU.
struct TnqJob {
    SegPtrs segs;
    int ycol0, Ov;
    float* partial;
};
template <int KT, int RC, bool PLANAR>
__global__ __launch_bounds__(256, 2) void gemm_tnq_pair_kernel(TnqJob ja, TnqJob jb, int nseg, int F, int R,
                                                              const float* __restrict__ dY, int ldy, int rows_per_split, int flags) {
    const int y = (int)blockIdx.y, split = y >> 1;
    const int which = (y & 1) ^ ((y >> 8) & 1);      // WG y lands on CU y % 256, slot y / 256: one job of each kind per CU
    if (which == 0)
        gemm_tnq_body<KT, 4, RC, false, PLANAR, false>(ja.segs, nseg, F, R, dY, ldy, ja.ycol0, ja.Ov, ja.partial, rows_per_split, 0, 0, 0, flags,
                                                       (int)blockIdx.x, split);
    else
        gemm_tnq_body<KT, 2, RC, false, PLANAR, false>(jb.segs, nseg, F, R, dY, ldy, jb.ycol0, jb.Ov, jb.partial, rows_per_split, 0, 0, 0, flags,
                                                       (int)blockIdx.x, split);
}

}  // namespace eeg
