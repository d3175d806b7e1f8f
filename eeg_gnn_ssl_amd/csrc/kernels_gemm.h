// fp32 MFMA GEMMs for the hoisted (non-recurrent) parts of the DCGRU layer.
//
//  gemm_nn_kernel:  C[R x O] = [A_0 | A_1 | ... ] [R x nseg*F] * B + bias     (x-part of the
//                   diffusion convolution for all T*B samples at once; also dY * W^T for dX)
//                   A is given as `nseg` row-major planes (R x F) = hop planes; B is the
//                   fragment-packed weight block (kernels_pack.h).
//  gemm_tn_kernel:  P[split][nseg*F x O] = sum over a row range of A^T dY      (weight grads;
//                   split-K partials are reduced in fixed order by reduce_unpack_kernel).
//
// Both use v_mfma_f32_16x16x4_f32 on LDS-staged, double-buffered tiles with one barrier per chunk.
// The shipped variants (gemm_nn_dma_kernel, gemm_tn_dma_kernel) stage with LDS-DMA
// (global_load_lds_dwordx4); the register-staged gemm_nn_kernel / gemm_tn_kernel remain for shapes the
// DMA layouts do not cover (odd leading dimensions, O <= 32, K chunks that are not multiples of 16/20).
// Roofline: fp32 MFMA (157.3 TFLOP/s), see DESIGN.md.
#pragma once
#include "common.h"

namespace eeg {

struct SegPtrs { const float* p[kMaxM]; };

// ---------------------------------------------------------------------------------------------
// NN: workgroup tile = 128 rows x (2*NCTW*16) cols, 4 waves as 2 (rows) x 2 (cols); each wave
// 4 row tiles x NCTW col tiles.  K is walked segment by segment in chunks of KC (F % KC == 0).
// LDS per buffer: A [128][KCS] + B [KC/4][NB][64], NB = 2*NCTW col tiles of this block.
template <int NCTW, int KC>
__global__ __launch_bounds__(256) void gemm_nn_kernel(SegPtrs segs, int nseg, int F, int R,
                                                      const float* __restrict__ Bp, int nct_total,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ C, int ldc, int O) {
    constexpr int KCS = lds_stride(KC), NB = 2 * NCTW, KSC = KC / 4;
    constexpr int A_FLOATS = 128 * KCS, B_FLOATS = KSC * NB * 64;
    constexpr int A_LD = (128 * KC / 4 + 255) / 256;        // float4 loads per thread per chunk
    constexpr int B_LD = (B_FLOATS / 4 + 255) / 256;
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
    const int row0 = blockIdx.x * 128, ct0 = blockIdx.y * NB;
    const int nchunk_seg = F / KC, nchunks = nseg * nchunk_seg;

    f32x4 acc[4][NCTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NCTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[A_LD], rb[B_LD];
    auto gload = [&](int chunk) {
        const int seg = chunk / nchunk_seg, kc0 = (chunk % nchunk_seg) * KC;
        const float* A = segs.p[seg];
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int q = tid + 256 * i, row = q / (KC / 4), c4 = q % (KC / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < 128 * KC / 4 && row0 + row < R)
                v = *reinterpret_cast<const float4*>(A + (size_t)(row0 + row) * F + kc0 + 4 * c4);
            ra[i] = v;
        }
        const int gks0 = (seg * F + kc0) / 4;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int q = tid + 256 * i;                 // float4 index inside [KSC][NB][64]
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < B_FLOATS / 4) {
                const int ks = q / (NB * 16), rem = q % (NB * 16), ct = rem / 16, l4 = rem % 16;
                if (ct0 + ct < nct_total)
                    v = *reinterpret_cast<const float4*>(Bp + ((size_t)(gks0 + ks) * nct_total + ct0 + ct) * 64 + 4 * l4);
            }
            rb[i] = v;
        }
    };
    auto lstore = [&](int buf) {
        float* As = sm + buf * (A_FLOATS + B_FLOATS);
        float* Bs = As + A_FLOATS;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int q = tid + 256 * i, row = q / (KC / 4), c4 = q % (KC / 4);
            if (q < 128 * KC / 4) {
                float* d = As + row * KCS + 4 * c4;
                d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int q = tid + 256 * i;
            if (q < B_FLOATS / 4) *reinterpret_cast<float4*>(Bs + 4 * q) = rb[i];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) gload(ch + 1);
        const float* As = sm + buf * (A_FLOATS + B_FLOATS);
        const float* Bs = As + A_FLOATS;
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) {
            float a[4], b[NCTW];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[(wr * 64 + i * 16 + lr) * KCS + 4 * ks + lg];
#pragma unroll
            for (int j = 0; j < NCTW; ++j) b[j] = Bs[(ks * NB + wc * NCTW + j) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NCTW; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
        if (ch + 1 < nchunks) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NCTW; ++j) {
        const int col = (ct0 + wc * NCTW + j) * 16 + lr;
        const float bv = (bias != nullptr && col < O) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wr * 64 + i * 16 + 4 * lg + r;
                if (row < R && col < O) C[(size_t)row * ldc + col] = acc[i][j][r] + bv;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// NN with LDS-DMA staging (the shipped variant): same 128 x (2*NCTW*16) workgroup tile as above, but
//  (i) MFMAs are issued transposed (weights as A operand), so a lane owns 4 consecutive output columns of
//      one row -> 16-byte bias loads / C stores;
//  (ii) the K chunks go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no
//      ds_write pass, no vmcnt wait in front of it; 2-3 workgroups share a CU.
// The LDS image of a DMA is lane-linear, so the A tile is stored unpadded, [128 rows][KC floats]:
//   KC = 20: row stride 20 floats -> the 16 rows x 4 k-lanes of a fragment read fall on 64 distinct banks;
//   KC = 16: row stride 16 would be 4-way conflicted, so the 16-byte pieces of a row are XOR-swizzled
//            by ((row >> 2) & 3) on the SOURCE side (which piece a lane fetches) and on the READ side.
// The B chunk is already fragment-ordered in the pack and is copied verbatim.
// Rows >= R are clamped to R-1 (fetched, never stored); requires F % KC == 0, ldc % 4 == 0.
template <int NCTW, int KC, int MINB>
__global__ __launch_bounds__(256, MINB) void gemm_nn_dma_kernel(SegPtrs segs, int nseg, int F, int R,
                                                                const float* __restrict__ Bp, int nct_total,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ C, int ldc, int O,
                                                                int btT, int btB, int btN) {
    constexpr int NB = 2 * NCTW, KSC = KC / 4, Q = KC / 4;              // Q = 16-byte pieces per A row
    constexpr int A_FLOATS = 128 * KC, B_FLOATS = KSC * NB * 64;
    constexpr int A_INS = A_FLOATS / 256, B_INS = B_FLOATS / 256, INS = A_INS + B_INS;   // wave-DMAs per chunk
    constexpr int NI = (INS + 3) / 4;                                   // per wave
    static_assert(A_FLOATS % 256 == 0 && B_FLOATS % 256 == 0, "chunk must be whole wave-DMAs");
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
    const int row0 = blockIdx.x * 128, ct0 = blockIdx.y * NB;
    const int nchunk_seg = F / KC, nchunks = nseg * nchunk_seg;

    // per-lane source offsets of this wave's DMAs (chunk-independent part), in floats
    // btT > 0: the A segments are BATCH-major (btB clips x btT steps x btN nodes x F): the time-major row
    // r = (t*B + b)*N + n that C uses lives at row (b*T + t)*N + n of every segment (the model input as the trainer
    // holds it, model.py:253, and its hop planes in the same order -- no time-major copy is made)
    unsigned src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = wave + 4 * i;                                     // DMA index inside the chunk
        if (j < A_INS) {
            const int s4 = j * 64 + lane, row = s4 / Q, piece = s4 % Q;
            int grow = row0 + row < R ? row0 + row : R - 1;
            const int c4 = Q == 4 ? (piece ^ ((row >> 2) & 3)) : piece;
            if (btT > 0) {
                const int sm_ = grow / btN, n = grow - sm_ * btN, t = sm_ / btB, b = sm_ - t * btB;
                grow = (b * btT + t) * btN + n;
            }
            src[i] = (unsigned)grow * F + 4 * c4;
        } else {
            const int s4 = (j - A_INS) * 64 + lane;                     // float4 index in [KSC][NB][16]
            const int ks = s4 / (NB * 16), rem = s4 % (NB * 16), ct = rem / 16, l4 = rem % 16;
            const int gct = ct0 + ct < nct_total ? ct0 + ct : nct_total - 1;
            src[i] = ((unsigned)ks * nct_total + gct) * 64 + 4 * l4;
        }
    }
    auto dma = [&](int chunk, int buf) {
        const int seg = chunk / nchunk_seg, kc0 = (chunk % nchunk_seg) * KC;
        const float* Ab = segs.p[seg] + kc0;
        const float* Bb = Bp + (size_t)((seg * F + kc0) / 4) * nct_total * 64;
        float* base = sm + buf * (A_FLOATS + B_FLOATS);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = wave + 4 * i;
            if (j < INS) lds_dma16(base + j * 256, (j < A_INS ? Ab : Bb) + src[i]);
        }
    };

    f32x4 acc[4][NCTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NCTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment read offsets: A[row][4*ks + lg]
    int arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) arow[i] = (wr * 64 + i * 16 + lr) * KC;
    const int aswz = Q == 4 ? ((lr >> 2) & 3) : 0;                      // (row >> 2) & 3 with row = 16*x + lr
    auto compute = [&](int buf) {
        const float* As = sm + buf * (A_FLOATS + B_FLOATS);
        const float* Bs = As + A_FLOATS;
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) {
            float a[4], b[NCTW];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[arow[i] + 4 * (Q == 4 ? (ks ^ aswz) : ks) + lg];
#pragma unroll
            for (int j = 0; j < NCTW; ++j) b[j] = Bs[(ks * NB + wc * NCTW + j) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NCTW; ++j) acc[i][j] = mfma16(b[j], a[i], acc[i][j]);   // transposed
        }
    };
    dma(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) dma(ch + 1, buf ^ 1);                     // lands while this chunk is multiplied
        compute(buf);
        __syncthreads();                                                // (the compiler drains the DMA queue here)
    }
    float4 bv[NCTW];
#pragma unroll
    for (int j = 0; j < NCTW; ++j) {
        const int col = (ct0 + wc * NCTW + j) * 16 + 4 * lg;
        bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias != nullptr && col + 3 < O) bv[j] = *reinterpret_cast<const float4*>(bias + col);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + wr * 64 + i * 16 + lr;
        if (row >= R) continue;
#pragma unroll
        for (int j = 0; j < NCTW; ++j) {
            const int col = (ct0 + wc * NCTW + j) * 16 + 4 * lg;
            float* c = C + (size_t)row * ldc + col;
            if (col + 3 < O) {
                *reinterpret_cast<float4*>(c) = make_float4(acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y,
                                                            acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r < O) c[r] = acc[i][j][r] + (bias != nullptr ? bias[col + r] : 0.f);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// TN: workgroup output tile = 64 k-rows (one 64-wide feature block of one hop plane) x
// (2*NCTW*16) columns of dY; 4 waves as 2 (k) x 2 (cols), each 2 k-tiles x NCTW col tiles.
// The reduction runs over rows [split*rows_per_split, ...) in chunks of 32 rows.
// grid = (nseg * ceil(F/64), nsplit).  partial: [nsplit][nseg*F][Ov]; Ov <= 2*NCTW*16 valid
// columns (Ov % 4 == 0), the rest of the tile is zero-filled / not stored.
template <int NCTW>
__global__ __launch_bounds__(256) void gemm_tn_kernel(SegPtrs segs, int nseg, int F, int R,
                                                      const float* __restrict__ dY, int ldy, int ycol0, int Ov,
                                                      float* __restrict__ partial, int rows_per_split, int xcd_remap) {
    constexpr int RC = 32, O = 2 * NCTW * 16;
    constexpr int AS = 80;                                  // 64 + 16: stride % 32 == 16
    constexpr int YS = O + ((16 - (O % 32)) + 32) % 32;     // stride % 32 == 16
    constexpr int A_FLOATS = RC * AS, Y_FLOATS = RC * YS;
    constexpr int A_LD = RC * 64 / 4 / 256;                 // = 2
    constexpr int Y_LD = (RC * O / 4 + 255) / 256;
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
    const int nfb = ceil_div(F, 64);
    // Optional XCD-aware placement (off by default, see api.cpp: it measured slower): workgroups are
    // dealt round-robin to the 8 XCDs by linear id and every XCD has its own L2; with the remap all
    // k-blocks of one row split (they read the same dY rows) get linear ids congruent mod 8, i.e.
    // the same XCD.  Correctness does not depend on the placement.
    int kblock = blockIdx.x, split = blockIdx.y;
    if (xcd_remap) {                                       // host guarantees gridDim.y % 8 == 0
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, slot = lin >> 3;
        split = xcd + 8 * (slot / (int)gridDim.x);
        kblock = slot % (int)gridDim.x;
    }
    const int seg = kblock / nfb, f0 = (kblock % nfb) * 64;
    const int rbeg = split * rows_per_split;
    const int rend = (rbeg + rows_per_split < R) ? rbeg + rows_per_split : R;
    const float* A = segs.p[seg];

    f32x4 acc[2][NCTW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NCTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[A_LD], ry[Y_LD];
    auto gload = [&](int r0) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int q = tid + 256 * i, row = q / 16, c4 = q % 16;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + row < rend && f0 + 4 * c4 < F)
                v = *reinterpret_cast<const float4*>(A + (size_t)(r0 + row) * F + f0 + 4 * c4);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < Y_LD; ++i) {
            const int q = tid + 256 * i, row = q / (O / 4), c4 = q % (O / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < RC * O / 4 && r0 + row < rend && 4 * c4 < Ov)
                v = *reinterpret_cast<const float4*>(dY + (size_t)(r0 + row) * ldy + ycol0 + 4 * c4);
            ry[i] = v;
        }
    };
    auto lstore = [&](int buf) {
        float* At = sm + buf * (A_FLOATS + Y_FLOATS);
        float* Ys = At + A_FLOATS;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int q = tid + 256 * i, row = q / 16, c4 = q % 16;
            *reinterpret_cast<float4*>(At + row * AS + 4 * c4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < Y_LD; ++i) {
            const int q = tid + 256 * i, row = q / (O / 4), c4 = q % (O / 4);
            if (q < RC * O / 4) *reinterpret_cast<float4*>(Ys + row * YS + 4 * c4) = ry[i];
        }
    };

    const int nchunks = rend > rbeg ? ceil_div(rend - rbeg, RC) : 0;
    if (nchunks > 0) {
        gload(rbeg);
        lstore(0);
    }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) gload(rbeg + (ch + 1) * RC);
        const float* At = sm + buf * (A_FLOATS + Y_FLOATS);
        const float* Ys = At + A_FLOATS;
#pragma unroll
        for (int ks = 0; ks < RC / 4; ++ks) {
            float a[2], b[NCTW];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = At[(4 * ks + lg) * AS + (wk * 2 + i) * 16 + lr];
#pragma unroll
            for (int j = 0; j < NCTW; ++j) b[j] = Ys[(4 * ks + lg) * YS + (wc * NCTW + j) * 16 + lr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NCTW; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
        if (ch + 1 < nchunks) lstore(buf ^ 1);
        __syncthreads();
    }
    const size_t Ktot = (size_t)nseg * F;
    float* out = partial + (size_t)split * Ktot * Ov;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = f0 + (wk * 2 + i) * 16 + 4 * lg + r;
            if (f < F) {
#pragma unroll
                for (int j = 0; j < NCTW; ++j) {
                    const int col = (wc * NCTW + j) * 16 + lr;
                    if (col < Ov) out[((size_t)seg * F + f) * Ov + col] = acc[i][j][r];
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// TN with LDS-DMA staging: split-K over rows like gemm_tn_kernel; a workgroup owns KBW = 32*KTW
// consecutive columns of the CONCATENATED K axis (k = seg*F + f, so a block may span hop planes:
// every lane fetches from its own plane pointer) x O columns of dY, in RC-row chunks that go
// global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write pass).
// LDS rows are unpadded (KBW and O floats); the transposed fragment reads (lane group g reads row
// 4*ks + g) would then all hit the same banks, so the 16-byte pieces of row r are XOR-swizzled by
// 4*(r & 3), on the source side of the DMA and on the read side.  Pieces past the valid columns
// and rows past R are clamped (fetched, finite, never used: their outputs are not stored / the
// rows are zeroed in LDS before the last chunk is multiplied).
// 4 waves = 2 (k) x 2 (cols), each KTW k-tiles x NCTW col tiles.  grid = (ceil(K / KBW), nsplit).
// Requires O % 64 == 0 (NCTW in {2,4,6}), F % 4 == 0, Ov % 4 == 0, rows_per_split % RC == 0.
// WK = waves along k: 2 (4 waves, 64-column k-block, the default) or 4 (8 waves, 128-column k-block: every dY chunk is
// multiplied with twice as many A columns, i.e. half the dY re-reads per flop, at 2 waves per SIMD and workgroup).
template <int KTW, int NCTW, int RC, int WK = 2>
__global__ __launch_bounds__(128 * WK) void gemm_tn_dma_kernel(SegPtrs segs, int nseg, int F, int R,
                                                          const float* __restrict__ dY, int ldy, int ycol0, int Ov,
                                                          float* __restrict__ partial, int rows_per_split,
                                                          int btT, int btB, int btN, int xcd_remap) {
    constexpr int NW = 2 * WK, KBW = 16 * KTW * WK, KQ = KBW / 4, O = 2 * NCTW * 16, OQ = O / 4;
    constexpr int A_FLOATS = RC * KBW, Y_FLOATS = RC * O;
    constexpr int A_INS = A_FLOATS / 256, Y_INS = Y_FLOATS / 256, INS = A_INS + Y_INS, NI = (INS + NW - 1) / NW;
    static_assert(A_FLOATS % 256 == 0 && Y_FLOATS % 256 == 0 && RC % 4 == 0, "chunk must be whole wave-DMAs");
    EEG_DYN_SMEM(sm);
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int wk = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
    // xcd_remap (default where gridDim.y % 8 == 0): workgroups go round-robin over the 8 XCDs by linear id; with the remap
    // all k-blocks of one row split (they read the same dY rows) land on ONE XCD, i.e. behind one L2 -- the re-reads stop
    // at that L2 instead of going out to the fabric (PMC FETCH_SIZE of the x-part GEMM 634 -> 263 MB-units per launch)
    int kblock = blockIdx.x, split = blockIdx.y;
    if (xcd_remap) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, slot = lin >> 3;
        split = xcd + 8 * (slot / (int)gridDim.x);
        kblock = slot % (int)gridDim.x;
    }
    const int K = nseg * F, k0 = kblock * KBW;
    const int rbeg = split * rows_per_split;
    const int rend = (rbeg + rows_per_split < R) ? rbeg + rows_per_split : R;

    // this wave's DMAs: tile row, source base (plane of the lane's k column / dY) and (clamped) column
    // btT > 0: the A segments are BATCH-major (see gemm_nn_dma_kernel): their lanes walk the rows of the split
    // through (b, t, n) counters instead of a linear row index (no division in the loop); dY stays time-major
    int drow[NI];
    const float* dsrc[NI];
    int dld[NI];
    int mb[NI], mt[NI], mn[NI];
    bool m0[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = wave + NW * i;
        m0[i] = false;
        mb[i] = mt[i] = mn[i] = 0;
        if (j < A_INS) {
            const int s4 = j * 64 + lane, row = s4 / KQ, piece = (s4 % KQ) ^ (4 * (row & 3));
            int k = k0 + 4 * piece;
            if (k >= K) k = K - 4;
            drow[i] = row;
            dsrc[i] = segs.p[k / F] + k % F;
            dld[i] = F;
            if (btT > 0) {
                const int r = rbeg + row, sm_ = r / btN;
                m0[i] = true;
                mn[i] = r - sm_ * btN;
                mt[i] = sm_ / btB;
                mb[i] = sm_ - mt[i] * btB;
            }
        } else {
            const int s4 = (j - A_INS) * 64 + lane, row = s4 / OQ, piece = (s4 % OQ) ^ (4 * (row & 3));
            drow[i] = row;
            dsrc[i] = dY + ycol0 + (4 * piece < Ov ? 4 * piece : Ov - 4);
            dld[i] = ldy;
        }
    }
    auto dma = [&](int r0, int buf) {
        float* base = sm + buf * (A_FLOATS + Y_FLOATS);
        const bool tail = r0 + RC > R;                      // only the last chunk of the last split
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = wave + NW * i;
            if (j >= INS) continue;
            int row = r0 + drow[i];
            if (m0[i]) {                                    // chunks are requested in row order: advance by RC rows per call
                const int mapped = (mb[i] * btT + mt[i]) * btN + mn[i];
                mn[i] += RC;
                while (mn[i] >= btN) {
                    mn[i] -= btN;
                    if (++mb[i] == btB) { mb[i] = 0; ++mt[i]; }
                }
                row = row < R ? mapped : R - 1;            // (the last row maps to itself)
            } else if (tail && row >= R) {
                row = R - 1;
            }
            lds_dma16(base + j * 256, dsrc[i] + (size_t)row * dld[i]);
        }
    };

    f32x4 acc[KTW][NCTW];
#pragma unroll
    for (int i = 0; i < KTW; ++i)
#pragma unroll
        for (int j = 0; j < NCTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // swizzled fragment columns: element (row 4*ks + lg, col 16*ct + lr) lives at piece (4*ct + lr/4) ^ (4*lg)
    int acol[KTW], ycol[NCTW];
#pragma unroll
    for (int i = 0; i < KTW; ++i) acol[i] = (((4 * (wk * KTW + i) + (lr >> 2)) ^ (4 * lg)) << 2) + (lr & 3);
#pragma unroll
    for (int j = 0; j < NCTW; ++j) ycol[j] = (((4 * (wc * NCTW + j) + (lr >> 2)) ^ (4 * lg)) << 2) + (lr & 3);

    const int nchunks = rend > rbeg ? ceil_div(rend - rbeg, RC) : 0;
    if (nchunks > 0) dma(rbeg, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1, r0 = rbeg + ch * RC;
        if (ch + 1 < nchunks) dma(r0 + RC, buf ^ 1);
        float* At = sm + buf * (A_FLOATS + Y_FLOATS);
        float* Ys = At + A_FLOATS;
        if (r0 + RC > rend) {                               // partial last chunk: rows past the end contribute zero
            const int valid = rend - r0;
            for (int e = tid; e < (RC - valid) * OQ; e += 64 * NW)
                *reinterpret_cast<float4*>(Ys + valid * O + 4 * e) = make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
        }
#pragma unroll
        for (int ks = 0; ks < RC / 4; ++ks) {
            float a[KTW], b[NCTW];
#pragma unroll
            for (int i = 0; i < KTW; ++i) a[i] = At[(4 * ks + lg) * KBW + acol[i]];
#pragma unroll
            for (int j = 0; j < NCTW; ++j) b[j] = Ys[(4 * ks + lg) * O + ycol[j]];
#pragma unroll
            for (int i = 0; i < KTW; ++i)
#pragma unroll
                for (int j = 0; j < NCTW; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* out = partial + (size_t)split * K * Ov;
#pragma unroll
    for (int i = 0; i < KTW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + (wk * KTW + i) * 16 + 4 * lg + r;
            if (k < K) {
#pragma unroll
                for (int j = 0; j < NCTW; ++j) {
                    const int col = (wc * NCTW + j) * 16 + lr;
                    if (col < Ov) out[(size_t)k * Ov + col] = acc[i][j][r];
                }
            }
        }
}

}  // namespace eeg
