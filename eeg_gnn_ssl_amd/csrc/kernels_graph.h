// Per-clip correlation graph -> dual random-walk supports, on the GPU.
//
// Reference (CPU, inside the DataLoader workers): dataloader_detection.py:258-307 (_get_indiv_graphs:
// normalised 'valid' cross-correlation of every electrode pair of the (N, T*D) clip = |cosine Gram|,
// diagonal 1), data_utils.py:174-200 (keep_topk, top_k = 3, directed), utils.py:220-230 +
// dataloader_detection.py:346-349 (S1 = (D^-1 A)^T, S2 = (D_in^-1 A^T)^T).
//
// This is the memory-bound variant of the path: each 456 kB clip is read once for ~2.3 MFLOP.
//  * corr_gram_kernel: grid (B, NS).  Workgroup (b, sp) accumulates the Gram of clip b over its
//    share of the time steps with fp32 MFMA: a time step is staged global -> LDS by LDS-DMA as one
//    contiguous stream, lane (i, g) reads the 16 bytes X[t][node i][16q+4g .. +3] from LDS and feeds
//    them as BOTH operands (G = X X^T), one float per MFMA; the K order inside a 16-feature chunk is
//    permuted, which a sum over all k does not care about.  Algorithmic bytes: 4*T*N*D per clip.
//  * corr_finish_kernel: one workgroup per clip: fixed-order sum of the NS partial Grams,
//    normalisation, |.|, diag = 1, top-k per row, S1/S2.
#pragma once
#include "common.h"

namespace eeg {

constexpr int kGramTile = 256;                 // one 16x16 MFMA accumulator tile, C layout (r*64 + lane)
constexpr int kGramFloats = 3 * kGramTile;     // tiles (0,0), (0,1), (1,1) of the padded 32x32 Gram

// NQ = number of 16-feature chunks.  Every wave owns the time steps t0, t0 + 4*NS, ... of its clip and a private LDS
// buffer of one time step (N*D floats rounded up to whole 1-KB wave-DMAs): the step travels global -> LDS as ONE
// contiguous stream of global_load_lds_dwordx4 (full 128-byte lines; the round-1 kernel fetched 64-byte row pieces
// straight into MFMA operands and reached 2.8 TB/s), the fragments X[t][node i][16q + 4g ..] are then read from LDS
// with ds_read_b128, the DMA of the wave's NEXT step is issued into the same buffer as soon as they are in registers,
// and the MFMAs of the current step run while it flies.  No workgroup barrier in the loop; ~16 waves per CU keep
// enough bytes in flight.
// REM4 (at most 20 nodes): the second node tile holds at most 4 real rows, so its two Gram tiles -- 2/3 of the matrix
// work for 3 of 19 nodes -- are computed with v_mfma_f32_4x4x1 (16 independent 4x4 outer products, a quarter of the
// cost of a 16x16x4): A operand = X[16 + (lane & 3)][k], B operand = the tile-0 fragment (-> rows 16.. x cols 0..15) or
// the same A register (-> rows 16.. x cols 16..); every lane group then holds the partial sum of ITS k quarter, and
// the four are added once, at the end, in a fixed order.
template <int NQ, bool REM4>
__global__ __launch_bounds__(256) void corr_gram_kernel(const float* __restrict__ X, int T, int N, int D,
                                                        float* __restrict__ part, int step_floats) {
    EEG_DYN_SMEM(sm);                          // [4 waves][step_floats] staging | reused as [4 waves][kGramFloats] at the end
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;
    const int b = blockIdx.x, sp = blockIdx.y, NS = gridDim.y;
    f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = {0.f, 0.f, 0.f, 0.f}, c11 = {0.f, 0.f, 0.f, 0.f};
    f32x4 c00b = c00, c01b = c00, c11b = c00;          // REM4: second chains (odd k of every pair), added at the end
    const int i1 = REM4 ? 16 + (i & 3) : 16 + i;          // second-tile row this lane feeds
    const bool has0 = i < N, has1 = i1 < N;
    float* buf = sm + wave * step_floats;
    const int nd = N * D, ndma = step_floats / 256;        // valid floats of a step; wave-DMAs per step
    const size_t total = (size_t)gridDim.x * T * nd;       // floats of X (the last DMA of the last step is clamped)
    auto stage = [&](int t) {
        const size_t base = ((size_t)b * T + t) * nd;
        for (int j = 0; j < ndma; ++j) {
            size_t off = base + (size_t)(j * 64 + lane) * 4;
            if (off + 4 > total) off = total - 4;           // past the end of X: any valid 16 bytes (never used)
            lds_dma16(buf + j * 256, X + off);
        }
    };
    const int r0 = (has0 ? i : 0) * D, r1 = (has1 ? i1 : 0) * D;
    const int t0 = sp * 4 + wave, dt = 4 * NS;
    if (t0 < T) stage(t0);
    for (int t = t0; t < T; t += dt) {
        float4 a0[NQ], a1[NQ];
        EEG_WAVE_SYNC();                                    // every lane's pieces of the step have been requested
#pragma unroll
        for (int q = 0; q < NQ; ++q) {                     // (the compiler waits for the DMA in front of the first read)
            const int f = 16 * q + 4 * g;                   // D % 4 == 0 (checked by the host)
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 v0 = *reinterpret_cast<const float4*>(buf + r0 + (f < D ? f : 0));
            const float4 v1 = *reinterpret_cast<const float4*>(buf + r1 + (f < D ? f : 0));
            a0[q] = (has0 && f < D) ? v0 : z;
            a1[q] = (has1 && f < D) ? v1 : z;
        }
        // Round 5 (a race found by the margin-aware graph check of bench.py: 1-2 % of the calls returned ONE clip with a wrong Gram):
        // the fragment reads above must have RETURNED before the DMA below may overwrite the buffer.  A wave-level sync orders
        // instruction issue, not completion: under LDS contention (16 waves per CU) this wave's queued ds_reads could still be
        // waiting when the next step's data -- L2-hot, through the texture path -- landed in the buffer.
        EEG_LDS_WAIT();
        EEG_WAVE_SYNC();
        if (t + dt < T) stage(t + dt);                      // next step of this wave: flies during the MFMAs below
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 x0 = a0[q], x1 = a1[q];
            if constexpr (REM4) {       // c01 / c11 registers hold [lane][r] = partial of G[16 + r][node of the lane] (see above)
                // the quad's 16x16x4 MFMAs on two alternating chains (a lone chain issues every 52 cycles, not 32), then its
                // 4x4x1 MFMAs as one run (a change of shape costs ~11 cycles per 4x4x1, up to ~43 per run; chain_lab.hip)
                c00 = mfma16(x0.x, x0.x, c00); c00b = mfma16(x0.y, x0.y, c00b);
                c00 = mfma16(x0.z, x0.z, c00); c00b = mfma16(x0.w, x0.w, c00b);
                c01 = mfma4(x1.x, x0.x, c01); c11 = mfma4(x1.x, x1.x, c11); c01b = mfma4(x1.y, x0.y, c01b); c11b = mfma4(x1.y, x1.y, c11b);
                c01 = mfma4(x1.z, x0.z, c01); c11 = mfma4(x1.z, x1.z, c11); c01b = mfma4(x1.w, x0.w, c01b); c11b = mfma4(x1.w, x1.w, c11b);
            } else {
                c00 = mfma16(x0.x, x0.x, c00); c01 = mfma16(x0.x, x1.x, c01); c11 = mfma16(x1.x, x1.x, c11);
                c00 = mfma16(x0.y, x0.y, c00); c01 = mfma16(x0.y, x1.y, c01); c11 = mfma16(x1.y, x1.y, c11);
                c00 = mfma16(x0.z, x0.z, c00); c01 = mfma16(x0.z, x1.z, c01); c11 = mfma16(x1.z, x1.z, c11);
                c00 = mfma16(x0.w, x0.w, c00); c01 = mfma16(x0.w, x1.w, c01); c11 = mfma16(x1.w, x1.w, c11);
            }
        }
    }
    if constexpr (REM4) { c00 += c00b; c01 += c01b; c11 += c11b; }
    __syncthreads();                                        // staging buffers are free: reuse for the partial Grams
    float* mine = sm + wave * kGramFloats;
    if constexpr (REM4) {
        // tiles (0,1) and (1,1) in C layout (row = 4*(l>>4) + r, col = l & 15 at [r*64 + l]) from the per-lane-group partials:
        //   G[n][16 + r]     = sum_g c01[lane (n, g)][r]      -> tile (0,1), row n, col r
        //   G[16 + r][16 + j] = sum_g c11[lane (j, g)][r]      -> tile (1,1), row r, col j   (lanes with i < 4)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mine[0 * kGramTile + r * 64 + lane] = c00[r];
            mine[1 * kGramTile + r * 64 + lane] = 0.f;
            mine[2 * kGramTile + r * 64 + lane] = 0.f;
        }
        for (int gg = 0; gg < 4; ++gg) {                    // fixed order: lane group 0, 1, 2, 3
            EEG_WAVE_SYNC();
            if (g == gg) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mine[1 * kGramTile + (i & 3) * 64 + 16 * (i >> 2) + r] += c01[r];
                    if (i < 4) mine[2 * kGramTile + r * 64 + i] += c11[r];
                }
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mine[0 * kGramTile + r * 64 + lane] = c00[r];
            mine[1 * kGramTile + r * 64 + lane] = c01[r];
            mine[2 * kGramTile + r * 64 + lane] = c11[r];
        }
    }
    __syncthreads();
    float* out = part + ((size_t)b * NS + sp) * kGramFloats;
    for (int e = threadIdx.x; e < kGramFloats; e += 256)
        out[e] = (sm[e] + sm[kGramFloats + e]) + (sm[2 * kGramFloats + e] + sm[3 * kGramFloats + e]);
}

// grid B, block 256.  LDS: G[32][33] | A[32][33] | rowsum[32] | colsum[32]
__global__ __launch_bounds__(256) void corr_finish_kernel(const float* __restrict__ part, int NS, int N, int top_k,
                                                          float* __restrict__ adj_out, float* __restrict__ S1,
                                                          float* __restrict__ S2) {
    EEG_DYN_SMEM(sm);
    constexpr int LS = 33;
    float* G = sm;
    float* A = sm + 32 * LS;
    float* rsum = A + 32 * LS;
    float* csum = rsum + 32;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < kGramFloats; e += 256) {
        float s = 0.f;
        for (int sp = 0; sp < NS; ++sp) s += part[((size_t)b * NS + sp) * kGramFloats + e];
        const int tile = e / kGramTile, r = (e % kGramTile) / 64, l = e % 64;
        const int row = 4 * (l >> 4) + r, col = l & 15;
        if (tile == 0) G[row * LS + col] = s;
        else if (tile == 1) { G[row * LS + 16 + col] = s; G[(16 + col) * LS + row] = s; }
        else G[(16 + row) * LS + 16 + col] = s;
    }
    __syncthreads();
    // normalised cross-correlation at lag 0 (data_utils.py:203-222), abs, unit diagonal
    for (int e = tid; e < N * N; e += 256) {
        const int i = e / N, j = e % N;
        float v = G[i * LS + j];
        const float cxx = G[i * LS + i], cyy = G[j * LS + j];
        if (cxx != 0.f && cyy != 0.f) v = v / sqrtf(cxx * cyy);
        A[i * LS + j] = i == j ? 1.f : fabsf(v);
    }
    __syncthreads();
    // keep_topk(top_k, directed): row i keeps its diagonal and its top_k largest off-diagonal entries
    if (tid < N) {
        const int i = tid;
        unsigned keep = 1u << i;
        for (int k = 0; k < top_k; ++k) {
            int best = -1;
            float bv = -1.f;
            for (int j = 0; j < N; ++j) {
                if ((keep >> j) & 1u) continue;
                const float v = A[i * LS + j];
                if (v > bv) { bv = v; best = j; }
            }
            if (best >= 0) keep |= 1u << best;
        }
        float rs = 0.f;
        for (int j = 0; j < N; ++j) {
            const float v = ((keep >> j) & 1u) ? A[i * LS + j] : 0.f;
            A[i * LS + j] = v;
            rs += v;
        }
        rsum[i] = rs;
    }
    __syncthreads();
    if (tid < N) {
        float cs = 0.f;
        for (int i = 0; i < N; ++i) cs += A[i * LS + tid];
        csum[tid] = cs;
    }
    __syncthreads();
    // random-walk supports (utils.py:220-230: D^-1 A with 1/0 -> 0), transposed as the dataloader does
    for (int e = tid; e < N * N; e += 256) {
        const int i = e / N, j = e % N;
        const size_t o = (size_t)b * N * N + e;
        const float rinv = rsum[j] != 0.f ? 1.0f / rsum[j] : 0.f;
        const float cinv = csum[j] != 0.f ? 1.0f / csum[j] : 0.f;
        if (adj_out != nullptr) adj_out[o] = A[i * LS + j];
        S1[o] = rinv * A[j * LS + i];          // (D^-1 A)^T
        S2[o] = cinv * A[i * LS + j];          // (D_in^-1 A^T)^T
    }
}

}  // namespace eeg
