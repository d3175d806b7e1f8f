// OPT-IN three-term bf16 split of the hoisted NN GEMMs (eeg_layer_dims.pack3 != NULL; never the default, never the headline:
// the contract's arithmetic is the fp32 matrix pipe).  C[R x O] = [A_0 | A_1 | ...] W + bias with fp32 operands and fp32 results:
//     a = a_hi + a_mid + a_lo,   w = w_hi + w_mid + w_lo        (each term a bf16: together 24 mantissa bits)
//     a w ~= a_hi w_hi + (a_hi w_mid + a_mid w_hi) + (a_mid w_mid + a_hi w_lo + a_lo w_hi)            (6 of the 9 products)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation, smallest terms first.  Round 3 measured the lab version of this kernel
// (tools/micro/bf16x3_lab.hip) at 1.4x the fp32-MFMA kernel with the same error against an fp64 sum (2.2e-6 vs 2.7e-6 on values
// up to 5.5); this is that kernel with what the product needs: hop-plane segments, the batch-major row map of the model input,
// feature widths that are not multiples of 32 (F = 100: the last 32-chunk of a plane is masked on the A side and zero in the
// pack), bias, 12- or 10-tile column blocks (O = 192; M*Fin = 320 at M = 5).
// Reference semantics: model/cell.py:98-117 (the dense contraction of the diffusion convolution, x-part) and its input gradient.
#pragma once
#include "kernels_gemm.h"

namespace eeg {

__host__ __device__ inline unsigned short bf16_rne(float x) {          // round to nearest even (finite values), = v_cvt_pk_bf16_f32
    unsigned a = __builtin_bit_cast(unsigned, x);
    a += 0x7fffu + ((a >> 16) & 1u);
    return (unsigned short)(a >> 16);
}
__host__ __device__ inline float bf16_to_f32(unsigned short h) {
    return __builtin_bit_cast(float, (unsigned)h << 16);
}

// ---- the term packs of one cell ------------------------------------------------------------------------------------------
// [xw | dx], 16-bit elements.  A right-hand side of K rows (nseg planes of F rows, each padded to whole 32-chunks) and 16*nct
// columns is stored as [term 0..2][chunk][column tile][lane][8]: lane l of (chunk c, tile ct) holds rows 32*cc + 8*(l>>4) + i
// (i = 0..7) of plane c / nchp, column 16*ct + (l & 15) -- one B fragment (the first MFMA operand: transposed issue, like the
// fp32 kernels) per 16 bytes.
struct Pack3 {
    int xw_nchp, xw_nch, xw_nct;      // x-part:  nseg = M planes of Fin rows, 3H columns
    int dx_nch, dx_nct;               // dX:      one segment of 3H rows (3H % 32 == 0), round_up(M*Fin, 16) columns
    size_t xw, dx, total;             // offsets / size in 16-bit elements
};
__host__ __device__ inline Pack3 make_pack3(int Fin, int H, int M) {
    Pack3 p;
    p.xw_nchp = ceil_div(Fin, 32);
    p.xw_nch = M * p.xw_nchp;
    p.xw_nct = 3 * H / 16;
    p.dx_nch = 3 * H / 32;
    p.dx_nct = round_up(M * Fin, 16) / 16;
    p.xw = 0;
    p.dx = (size_t)3 * p.xw_nch * p.xw_nct * 512;
    p.total = p.dx + (size_t)3 * p.dx_nch * p.dx_nct * 512;
    return p;
}
__host__ __device__ inline bool pack3_supported(int Fin, int H, int M) { return H == 64 && Fin % 4 == 0 && M >= 1 && M <= kMaxM; }

// Wg ((Fin+H)*M, 2H), Wc ((Fin+H)*M, H): reference layout, row = f*M + m (cell.py:98-116)
__global__ void pack_cell_bf3_kernel(const float* __restrict__ Wg, const float* __restrict__ Wc, int Fin, int H, int M,
                                     unsigned short* __restrict__ out) {
    const Pack3 p = make_pack3(Fin, H, M);
    const size_t n_xw = (size_t)p.xw_nch * p.xw_nct * 512, n_dx = (size_t)p.dx_nch * p.dx_nct * 512;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_xw + n_dx; idx += stride) {
        const bool dx = idx >= n_xw;
        const size_t e = dx ? idx - n_xw : idx;
        const int i = e & 7, lane = (e >> 3) & 63, nct = dx ? p.dx_nct : p.xw_nct;
        const int ct = (int)((e >> 9) % nct), c = (int)((e >> 9) / nct);
        float v = 0.f;
        if (!dx) {                     // W^x[k = (m, f)][j]
            const int m = c / p.xw_nchp, f = 32 * (c % p.xw_nchp) + 8 * (lane >> 4) + i, j = 16 * ct + (lane & 15);
            if (f < Fin) v = j < 2 * H ? Wg[((size_t)f * M + m) * (2 * H) + j] : Wc[((size_t)f * M + m) * H + (j - 2 * H)];
        } else {                       // (W^x)^T[k = o][j = m*Fin + f]
            const int o = 32 * c + 8 * (lane >> 4) + i, j = 16 * ct + (lane & 15);
            if (j < M * Fin) {
                const int m = j / Fin, f = j % Fin;
                v = o < 2 * H ? Wg[((size_t)f * M + m) * (2 * H) + o] : Wc[((size_t)f * M + m) * H + (o - 2 * H)];
            }
        }
        const unsigned short h = bf16_rne(v);
        const float r1 = v - bf16_to_f32(h);
        const unsigned short md = bf16_rne(r1);
        const unsigned short lo = bf16_rne(r1 - bf16_to_f32(md));
        unsigned short* base = out + (dx ? p.dx : p.xw);
        const size_t ts = dx ? n_dx : n_xw;
        base[e] = h;
        base[ts + e] = md;
        base[2 * ts + e] = lo;
    }
}

// x[0..7] -> three fragments (hi, mid, lo)
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    u32x4 H, M, L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned ph = pk_bf16(a, b);
        const float ra = a - __builtin_bit_cast(float, ph << 16), rb = b - __builtin_bit_cast(float, ph & 0xffff0000u);
        const unsigned pm = pk_bf16(ra, rb);
        const float sa = ra - __builtin_bit_cast(float, pm << 16), sb = rb - __builtin_bit_cast(float, pm & 0xffff0000u);
        H[i] = ph; M[i] = pm; L[i] = pk_bf16(sa, sb);
    }
    h = __builtin_bit_cast(bf16x8, H); m = __builtin_bit_cast(bf16x8, M); l = __builtin_bit_cast(bf16x8, L);
}

// One workgroup = 128 rows x NTB column tiles; wave w: rows 32w .. 32w+31 (two 16-row tiles) x NTB tiles.  The three term blocks
// of a chunk (3 x NTB KB) are staged in LDS (two stages) and shared by the four waves; the activations come straight from global
// memory (32 bytes per lane and row tile) and are split on the fly.  grid (ceil(R / 128), nct_total / NTB).
template <int NTB>
__global__ __launch_bounds__(256, 2) void gemm_nn_bf3_kernel(SegPtrs segs, int nseg, int F, int R,
                                                             const unsigned short* __restrict__ Wp, int nct_total,
                                                             const float* __restrict__ bias, float* __restrict__ C, int ldc, int O,
                                                             int btT, int btB, int btN) {
    static_assert(NTB % 2 == 0, "column tiles are processed in pairs");
    EEG_DYN_SMEM(smf);                                                         // 2 stages x 3 terms x NTB KB
    unsigned char* smem = reinterpret_cast<unsigned char*>(smf);
    constexpr int TERM_B = NTB * 1024, STAGE_B = 3 * TERM_B, PIECES = 3 * NTB * 64, NPB = (PIECES + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const int row0 = blockIdx.x * 128 + 32 * w, ct0 = blockIdx.y * NTB;
    const int nchp = ceil_div(F, 32), nch = nseg * nchp;
    const size_t term_stride = (size_t)nch * nct_total * 512;                  // 16-bit elements per term
    f32x4 acc[2][NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = 16 * (ct0 + j) + 4 * lg + r;
                b4[r] = col < O ? bias[col] : 0.f;
            }
        }
        acc[0][j] = b4;
        acc[1][j] = b4;
    }
    // rows of this lane's two row tiles in storage order (batch-major input: the (b, t) row map of gemm_nn_dma_kernel)
    size_t arow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = row0 + 16 * i + lr;
        if (r >= R) r = R - 1;
        if (btT > 0) {
            const int sm_ = r / btN, n = r - sm_ * btN, t = sm_ / btB, b = sm_ - t * btB;
            r = (b * btT + t) * btN + n;
        }
        arow[i] = (size_t)r * F;
    }
    u32x4 pb[NPB];                                                             // the next chunk's term blocks, in flight
    auto fetch_b = [&](int c) {
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            const int e = tid + 256 * q;
            if (e < PIECES) {
                const int t = e / (NTB * 64), r = e % (NTB * 64);
                pb[q] = *reinterpret_cast<const u32x4*>(Wp + t * term_stride + ((size_t)c * nct_total + ct0) * 512 + (size_t)r * 8);
            }
        }
    };
    auto put_b = [&](int st) {
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            const int e = tid + 256 * q;
            if (e < PIECES) *reinterpret_cast<u32x4*>(smem + (size_t)st * STAGE_B + (size_t)e * 16) = pb[q];
        }
    };
    float xa[2][8];
    auto load_a = [&](int c) {
        const int seg = c / nchp, f0 = 32 * (c - seg * nchp) + 8 * lg;
        const float* A = segs.p[seg];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (f0 < F) v0 = *reinterpret_cast<const f32x4*>(A + arow[i] + f0);          // F % 4 == 0: a 16-byte piece is all in or all out
            if (f0 + 4 < F) v1 = *reinterpret_cast<const f32x4*>(A + arow[i] + f0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xa[i][e] = v0[e]; xa[i][4 + e] = v1[e]; }
        }
    };
    fetch_b(0);
    load_a(0);
    put_b(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int st = c & 1;
        bf16x8 ah[2], am[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) split3(xa[i], ah[i], am[i], al[i]);
        if (c + 1 < nch) { fetch_b(c + 1); load_a(c + 1); }                   // in flight during the MFMAs below
        const unsigned char* sb = smem + (size_t)st * STAGE_B + (size_t)lane * 16;
        // two column tiles at a time and the partial products outermost: consecutive MFMAs go to four different accumulators
        // (six products into one accumulator back to back would be one dependent chain); smallest terms first
#pragma unroll
        for (int j = 0; j < NTB; j += 2) {
            bf16x8 bh[2], bm[2], bl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bh[u] = *reinterpret_cast<const bf16x8*>(sb + (j + u) * 1024);
                bm[u] = *reinterpret_cast<const bf16x8*>(sb + TERM_B + (j + u) * 1024);
                bl[u] = *reinterpret_cast<const bf16x8*>(sb + 2 * TERM_B + (j + u) * 1024);
            }
#define EEG_BF_TERM(B, Aop)                                                                      \
            _Pragma("unroll") for (int u = 0; u < 2; ++u)                                        \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                    \
                    acc[i][j + u] = mfma_bf16(B[u], Aop[i], acc[i][j + u]);
            EEG_BF_TERM(bh, al)
            EEG_BF_TERM(bl, ah)
            EEG_BF_TERM(bm, am)
            EEG_BF_TERM(bh, am)
            EEG_BF_TERM(bm, ah)
            EEG_BF_TERM(bh, ah)
#undef EEG_BF_TERM
        }
        if (c + 1 < nch) put_b(st ^ 1);                                       // (stage st^1 was last read in iteration c-1: behind its barrier)
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = row0 + 16 * i + lr;
        if (r < R) {
#pragma unroll
            for (int j = 0; j < NTB; ++j) {
                const int col = 16 * (ct0 + j) + 4 * lg;
                if (col + 3 < O) *reinterpret_cast<f32x4*>(C + (size_t)r * ldc + col) = acc[i][j];
                else
                    for (int e = 0; e < 4; ++e)
                        if (col + e < O) C[(size_t)r * ldc + col + e] = acc[i][j][e];
            }
        }
    }
}

}  // namespace eeg
