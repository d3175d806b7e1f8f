// (shared declarations; the kernels are in kernels_spectral.h, instantiated in spec_inst.cpp)
// Spectral form of the hoisted x-part of the diffusion convolution for ONE SHARED SYMMETRIC support (the scaled Laplacian of the
// distance graph: filter_type "laplacian", cell.py:151-158 / utils.calculate_scaled_laplacian).
//
// All hop matrices of such a cell are Chebyshev polynomials of one symmetric S = U diag(lam) U^T (cell.py:83-93: x1 = S x0,
// x2 = 2 S x1 - x0, ... => P_m = T_m(S) = U T_m(lam) U^T), so
//     sum_m P_m X W_m  =  U * [ xh_i * Wt_i ]_i        with  Xh = U^T X  (node mix of the layer input),
//                                                            Wt_i = sum_m T_m(lam_i) W_m  (one (Fin x 3H) matrix per graph frequency i)
// i.e. the hoisted GEMM contracts over K = Fin instead of K = M * Fin (a third of the MFMA work at K = 2), at the price of two
// HBM-bound node mixes (U^T on the Fin-wide input, U on the 3H-wide pre-activations).  The backward mirrors it:
//     dYh = U^T dXW,   dWt_i = Xh_i^T dYh_i,   dW_m = sum_i T_m(lam_i) dWt_i,   dX = U * [ dYh_i Wt_i^T ]_i.
// Only the hoisted x-part changes; the recurrence (h-part) keeps the hop polynomials.  Exact up to re-association.
//
// Layouts: the spectral-side tensors are NODE-major, (N, Sp, F) with Sp = round_up(S, 16) rows per graph frequency (pad rows are
// written as zeros), so that each frequency is a plain (Sp x F) matrix of the grouped GEMMs (kernels_gemm_g.h).
//
//  * spectral_basis_kernel: U, T_m(lam) and a residual from the support (parallel-order two-sided Jacobi in fp64, one workgroup).
//  * pack_spectral_kernel:  Wt_i (and Wt_i^T for dX) in the quad order of the grouped NN GEMM.
//  * spec_mix_in_kernel / spec_mix_out_kernel: the HBM-bound node mixes (one thread = one 16-byte feature column of one sample,
//    coefficients wave-uniform scalars -- the construction of diffuse_fwd_stream_kernel).
//  * spec_fold_block: dW_m = sum_i T_m(lam_i) (sum over the row splits of frequency i), fixed order.
//  * kernels_gemm_f.h (second half of round 6): gemm_nnf_kernel (Yh_i = Xh_i Wt_i, weights in registers), gemm_tnf_kernel (the three
//    weight-gradient contractions in one pass over dYh), gemm_dxf_kernel (dX = U [dYh_i Wt_i^T]_i in one kernel); the grouped round-5
//    kernels of kernels_gemm_g.h stay as the path of the widths those do not instantiate.  The 3H-wide mixes (U Yh, U^T dXW) and
//    U^T h / U^T (r*h) run inside the recurrent kernels (kernels_seq.h, SPEC instantiations).
#pragma once
#include "common.h"
#include "nnq_order.h"

namespace eeg {

// ---- the basis block: U (N*N, row n = node, column i = frequency) | tc[kMaxM][32] = T_m(lam_i) | info[32] | csum[32] = sum_n U[n][i]
constexpr int kSpecTc = kMaxM * 32, kSpecInfo = 32, kSpecCsum = 32;
__host__ __device__ constexpr int spec_basis_floats(int N) { return N * N + kSpecTc + kSpecInfo + kSpecCsum; }
__host__ __device__ constexpr int spec_csum_offset(int N) { return N * N + kSpecTc + kSpecInfo; }
__host__ __device__ constexpr int spec_rows(int S) { return round_up(S, 16); }
constexpr int kSpecSweeps = 12;

// ---- per-frequency weight packs of one cell ------------------------------------------------------------------------------------
// sxq : N blocks, block i = Wt_i  (K = Fin) x (3H) in the quad order of gemm_nng_kernel (nnq order over one segment of width Fin)
// sxtq: N blocks, block i = Wt_i^T (K = 3H)  x round_up(Fin, 16) columns (zero beyond Fin), same order over one segment of width 3H
// sxr : N blocks, block i = Wt_i row-major, round_up(Fin, 8) rows (zero beyond Fin) x 3H columns: gemm_nnf_kernel (kernels_gemm_f.h)
struct SpecPack {
    int Fin, H, M, N;
    int nch_x, nct_x, nch_t, nct_t;
    size_t sxq, sxq_stride, sxtq, sxtq_stride, sxr, sxr_stride, total;
};
__host__ __device__ inline SpecPack make_spec_pack(int Fin, int H, int M, int N) {
    SpecPack p;
    p.Fin = Fin; p.H = H; p.M = M; p.N = N;
    p.nch_x = make_nnq_order(1, Fin).nch; p.nct_x = 3 * H / 16;
    p.nch_t = make_nnq_order(1, 3 * H).nch; p.nct_t = round_up(Fin, 16) / 16;
    p.sxq_stride = (size_t)p.nch_x * p.nct_x * 256;
    p.sxtq_stride = (size_t)p.nch_t * p.nct_t * 256;
    p.sxq = 0;
    p.sxtq = p.sxq + p.sxq_stride * N;
    p.sxr = p.sxtq + p.sxtq_stride * N;
    p.sxr_stride = (size_t)round_up(Fin, 8) * 3 * H;
    p.total = p.sxr + p.sxr_stride * N;
    return p;
}
// ---- weight-gradient fold ------------------------------------------------------------------------------------------------------------
// partial [N * spg][K][O]: split (i, ls) = row split ls of frequency i (fixed-order split-K partials of a grouped TN GEMM).
//   kind 0 (x-part,  K = Fin, O = 3H): dW rows f*M + m       (columns < 2H -> dWg, the rest -> dWc)
//   kind 1 (h-gate,  K = H,   O = 2H): dWg rows (Fin + f)*M + m
//   kind 2 (h-cand,  K = H,   O = H ): dWc rows (Fin + f)*M + m
// value = sum_i T_m(lam_i) * sum_ls partial[i*spg + ls][f][o].  Block = 16 split groups x 16 float4 columns like
// reduce_unpack_block (kernels_pack.h): group g walks the splits g, g+16, ... in order with M accumulators, the 16 group sums are
// added in group order.  LDS: [16][16] float4 per hop slot, one slot at a time.
struct SpecFoldJob {
    const float* part;      // nullptr: no job
    int spg, K, O, nblocks;
};
struct SpecFoldJobs {
    SpecFoldJob j[3];       // kinds 0, 1, 2
    const float* basis;
    int N;
};
// LDS: [16][16] float4 reduction tile, then the coefficient table tcl[m][sp] = T_m(lam_{sp / spg}) of this job (M * N * spg floats;
// spec_fold_lds_bytes): the split loop then is loads + multiply-adds only, eight loads in flight per thread.
__host__ __device__ inline size_t spec_fold_lds_bytes(int M, int nsplit_max) { return 256 * 16 + (size_t)M * nsplit_max * sizeof(float); }
template <int MM>
__device__ __forceinline__ void spec_fold_block_m(int block, const SpecFoldJob& jb, int kind, const float* __restrict__ basis, int N,
                                                   int acc_flag, int Fin, int H, int M, float* __restrict__ dWg, float* __restrict__ dWc) {
    EEG_DYN_SMEM(sm);
    float4 (*red)[16] = reinterpret_cast<float4 (*)[16]>(sm);
    float* tcl = sm + 256 * 4;
    const int O = jb.O;
    const size_t total = (size_t)jb.K * O;
    const int g = threadIdx.x >> 4, q = threadIdx.x & 15;
    const size_t idx4 = ((size_t)block * 16 + q) * 4;
    const float* tc = basis + N * N;
    const int nsplit = N * jb.spg;
    for (int e = threadIdx.x; e < M * nsplit; e += 256) {
        const int m = e / nsplit, sp = e - m * nsplit;
        tcl[e] = tc[m * 32 + sp / jb.spg];
    }
    __syncthreads();
    float4 a[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) a[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx4 < total) {
        constexpr int UN = 8;
        for (int base = g; base < nsplit; base += 16 * UN) {
            float4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int sp = base + 16 * u;
                v[u] = sp < nsplit ? *reinterpret_cast<const float4*>(jb.part + (size_t)sp * total + idx4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int sp = base + 16 * u < nsplit ? base + 16 * u : 0;      // (beyond the end: v = 0)
#pragma unroll
                for (int m = 0; m < MM; ++m) {
                    if (m < M) {
                        const float t = tcl[m * nsplit + sp];
                        a[m].x = fmaf(t, v[u].x, a[m].x); a[m].y = fmaf(t, v[u].y, a[m].y);
                        a[m].z = fmaf(t, v[u].z, a[m].z); a[m].w = fmaf(t, v[u].w, a[m].w);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        if (m >= M) break;
        __syncthreads();
        red[g][q] = a[m];
        __syncthreads();
        if (g == 0 && idx4 < total) {
            float4 sum = red[0][q];
#pragma unroll
            for (int u = 1; u < 16; ++u) {
                const float4 v = red[u][q];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const size_t idx = idx4 + e;
                const int f = (int)(idx / O), o = (int)(idx % O);
                float* dst;
                if (kind == 0) dst = o < 2 * H ? &dWg[((size_t)f * M + m) * (2 * H) + o] : &dWc[((size_t)f * M + m) * H + (o - 2 * H)];
                else if (kind == 1) dst = &dWg[((size_t)(Fin + f) * M + m) * (2 * H) + o];
                else dst = &dWc[((size_t)(Fin + f) * M + m) * H + o];
                *dst = acc_flag ? *dst + sv[e] : sv[e];
            }
        }
    }
}
__device__ __forceinline__ void spec_fold_block(int block, const SpecFoldJob& jb, int kind, const float* __restrict__ basis, int N,
                                                int acc_flag, int Fin, int H, int M, float* __restrict__ dWg, float* __restrict__ dWc) {
    if (M <= 4) spec_fold_block_m<4>(block, jb, kind, basis, N, acc_flag, Fin, H, M, dWg, dWc);
    else spec_fold_block_m<kMaxM>(block, jb, kind, basis, N, acc_flag, Fin, H, M, dWg, dWc);
}

}  // namespace eeg
