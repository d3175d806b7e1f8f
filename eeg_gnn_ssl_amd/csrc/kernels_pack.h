// Weight (un)packing and hop-polynomial construction.
//
// Reference layout (model/cell.py:40-46, 98-116): dconv weight is ((Fin+H)*M, O) row-major with
// row = f*M + m (feature-major, hop-minor; f < Fin input features, f >= Fin hidden features);
// gate O = 2H (first H columns r, last H columns u), candidate O = H.
//
// Device layout: every GEMM B-operand is stored in MFMA-fragment order for
// v_mfma_f32_16x16x4_f32:  pack[(ks*NCT + ct)*64 + lane] = B[4*ks + (lane>>4)][16*ct + (lane&15)],
// with the K index hop-major (k = m*F + f) so that each hop plane is a contiguous K range.
// The four recurrent-kernel packs (bhg, bhc, b1, b2) use the quad-permuted K order `kperm`
// (common.h) instead of 4*ks + (lane>>4), matching their ds_read_b128 A-fragment reads.
#pragma once
#include "common.h"
#include "nnq_order.h"
#include "spec_common.h"
#include "pack_cell.h"

namespace eeg {

__global__ void pack_cell_kernel(const float* __restrict__ Wg, const float* __restrict__ bg,
                                 const float* __restrict__ Wc, const float* __restrict__ bc,
                                 float* __restrict__ out, CellPack p) {
    pack_cell_body(Wg, bg, Wc, bc, out, p, (int)blockIdx.x, (int)gridDim.x);
}

// One reference-layout dconv weight ((F*M), O) row = f*M+m  ->  fragment pack with hop-major K
// (k = m*F + f), NCT = O/16 column tiles (used by the stand-alone DiffusionGraphConv.forward).
__global__ void pack_dense_kernel(const float* __restrict__ W, int F, int M, int O, float* __restrict__ out) {
    const size_t total = (size_t)F * M * O;
    const int nct = O / 16;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int lane = e & 63, ct = (e >> 6) % nct, ks = (e >> 6) / nct;
        const int k = 4 * ks + (lane >> 4), j = 16 * ct + (lane & 15);
        const int m = k / F, f = k % F;
        out[e] = W[((size_t)f * M + m) * O + j];
    }
}

// The same weight transposed, for dX of the stand-alone dconv: right-hand side (K = O) x (J = M*F, hop-major,
// zero-padded to whole column tiles): B[o][m*F + f] = W[f*M + m][o].
__global__ void pack_dense_t_kernel(const float* __restrict__ W, int F, int M, int O, float* __restrict__ out) {
    const int nct = round_up(M * F, 16) / 16;
    const size_t total = (size_t)(O / 4) * nct * 64;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int lane = e & 63, ct = (e >> 6) % nct, ks = (e >> 6) / nct;
        const int o = 4 * ks + (lane >> 4), j = 16 * ct + (lane & 15);
        float v = 0.f;
        if (j < M * F) v = W[((size_t)(j % F) * M + j / F) * O + o];
        out[e] = v;
    }
}

// nn.Linear weight W (Out x In) -> fragment pack of a (K x O) right-hand side with zero-padded
// column tiles: transposed = 1: K = In, O = Out (y = x W^T);  transposed = 0: K = Out, O = In (dx = dy W).
__global__ void pack_linear_kernel(const float* __restrict__ W, int Out, int In, int transposed, float* __restrict__ out) {
    const int K = transposed ? In : Out, O = transposed ? Out : In;
    const int nct = (O + 15) / 16;
    const size_t total = (size_t)(K / 4) * nct * 64;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int lane = e & 63, ct = (e >> 6) % nct, ks = (e >> 6) / nct;
        const int k = 4 * ks + (lane >> 4), j = 16 * ct + (lane & 15);
        float v = 0.f;
        if (j < O) v = transposed ? W[(size_t)j * In + k] : W[(size_t)k * In + j];
        out[e] = v;
    }
}

// Sum split-K partials [nsplit][K][O] in fixed order (deterministic) and scatter into the
// reference-layout gradient tensors.  kind 0: x-part (K = M*Fin, O = 3H); 1: h-gate (K = M*H,
// O = 2H -> dWg rows Fin+f); 2: h-cand (K = M*H, O = H -> dWc rows Fin+f); 3: dWg = plain (K x O); 4: stand-alone dconv weight (K = M*F -> rows f*M+m).
// Block = 16 split groups x 16 float4 columns (64 consecutive elements of the K x O matrix): group g sums
// splits g, g+16, g+32, ... in that order (all its loads independent and in flight together -- the partials
// are read once from HBM, so the launch lives on memory-level parallelism), the 16 group sums are then added
// in group order.  Requires (K * O) % 4 == 0.
__device__ __forceinline__ void reduce_unpack_block(int block, const float* __restrict__ part, int nsplit, int K, int O,
                                                    int kind_flags, int Fin, int H, int M,
                                                    float* __restrict__ dWg, float* __restrict__ dWc) {
    EEG_DYN_SMEM(sm);                                 // [16 groups][16 float4]
    float4 (*red)[16] = reinterpret_cast<float4 (*)[16]>(sm);
    const size_t total = (size_t)K * O;
    const int g = threadIdx.x >> 4, q = threadIdx.x & 15;
    const size_t idx4 = ((size_t)block * 16 + q) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx4 < total) {
        for (int sp = g; sp < nsplit; sp += 16) {
            const float4 v = *reinterpret_cast<const float4*>(part + (size_t)sp * total + idx4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    red[g][q] = a;
    __syncthreads();
    if (g != 0 || idx4 >= total) return;
    float4 sum = red[0][q];
#pragma unroll
    for (int u = 1; u < 16; ++u) {
        const float4 v = red[u][q];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
    const bool accumulate = (kind_flags & 8) != 0;   // += into the gradient (shared decoder cell, model.py:126-143)
    const int kind = kind_flags & 7;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const size_t idx = idx4 + e;
        const int k = idx / O, o = idx % O;
        float* dst;
        if (kind == 0) {
            const int m = k / Fin, f = k % Fin;
            dst = o < 2 * H ? &dWg[((size_t)f * M + m) * (2 * H) + o] : &dWc[((size_t)f * M + m) * H + (o - 2 * H)];
        } else if (kind == 1) {
            dst = &dWg[((size_t)(Fin + k % H) * M + k / H) * (2 * H) + o];
        } else if (kind == 3) {                       // plain (K x O) matrix
            dst = &dWg[idx];
        } else if (kind == 4) {                       // stand-alone dconv: K = M*F hop-major -> row f*M + m of ((F*M) x O); F passed as Fin
            dst = &dWg[((size_t)(k % Fin) * M + k / Fin) * O + o];
        } else {
            dst = &dWc[((size_t)(Fin + k % H) * M + k / H) * H + o];
        }
        *dst = accumulate ? *dst + sv[e] : sv[e];
    }
}
__global__ __launch_bounds__(256) void reduce_unpack_kernel(const float* __restrict__ part, int nsplit, int K, int O,
                                                            int kind_flags, int Fin, int H, int M,
                                                            float* __restrict__ dWg, float* __restrict__ dWc) {
    reduce_unpack_block(blockIdx.x, part, nsplit, K, O, kind_flags, Fin, H, M, dWg, dWc);
}
// column sums of per-sample bias-gradient partials [B][3H] -> dbg (2H), dbc (H); fixed order.
// block = 256 threads = 16 columns x 16 row-slices; LDS combine in a fixed order (sm: 256 floats).
__device__ __forceinline__ void reduce_bias_block(int blk, float* sm, const float* __restrict__ part, int B, int H,
                                                  float* __restrict__ dbg, float* __restrict__ dbc) {
    const int c = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int j = blk * 16 + c;
    float s = 0.f;
    if (j < 3 * H)
        for (int b = q; b < B; b += 16) s += part[(size_t)b * 3 * H + j];
    sm[q * 16 + c] = s;
    __syncthreads();
    if (q == 0 && j < 3 * H) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += sm[i * 16 + c];
        if (j < 2 * H) dbg[j] = t; else dbc[j - 2 * H] = t;
    }
}
// The three weight-gradient GEMMs of one cell (x-part, h-gate, h-candidate: kinds 0, 1, 2) reduced by ONE launch:
// blocks [0, nb0) serve job 0, [nb0, nb0 + nb1) job 1, the rest job 2.
// Round 5: the bias gradients of the cell (column sums of the BPTT kernel's per-clip partials [B][3H]) ride in the same launch as a
// fourth job (blocks behind the three: one 5-us launch per layer and step less).
struct ReduceJobs {
    const float* part[3]; int nsplit[3]; int K[3]; int O[3]; int nblocks[3];
    const float* bias_part; int bias_B; float* dbg; float* dbc;              // bias_part == nullptr: no bias job
};
__global__ __launch_bounds__(256) void reduce_unpack3_kernel(ReduceJobs jobs, int flags, int Fin, int H, int M,
                                                             float* __restrict__ dWg, float* __restrict__ dWc) {
    int b = blockIdx.x, j = 0;
    if (b >= jobs.nblocks[0]) { b -= jobs.nblocks[0]; j = 1; }
    if (j == 1 && b >= jobs.nblocks[1]) { b -= jobs.nblocks[1]; j = 2; }
    if (j == 2 && b >= jobs.nblocks[2]) {            // the bias job (wave-uniform branch: whole blocks)
        EEG_DYN_SMEM(sm);
        reduce_bias_block(b - jobs.nblocks[2], sm, jobs.bias_part, jobs.bias_B, H, jobs.dbg, jobs.dbc);
        return;
    }
    reduce_unpack_block(b, jobs.part[j], jobs.nsplit[j], jobs.K[j], jobs.O[j], j | flags, Fin, H, M, dWg, dWc);
}

// The same launch for the spectral form (spec_common.h): jobs of kind 0..2 present in `sj` are folds of grouped TN partials
// (blocks first, in kind order); the h-part jobs NOT present there are the plain reductions of `jobs` (1, 2); the bias job last.
__global__ __launch_bounds__(256) void reduce_unpack3s_kernel(ReduceJobs jobs, SpecFoldJobs sj, int flags, int Fin, int H, int M,
                                                              float* __restrict__ dWg, float* __restrict__ dWc) {
    int b = blockIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (sj.j[k].part == nullptr) continue;
        if (b < sj.j[k].nblocks) {
            spec_fold_block(b, sj.j[k], k, sj.basis, sj.N, (flags & 8) != 0 ? 1 : 0, Fin, H, M, dWg, dWc);
            return;
        }
        b -= sj.j[k].nblocks;
    }
#pragma unroll
    for (int j = 1; j < 3; ++j) {
        if (sj.j[j].part != nullptr) continue;
        if (b < jobs.nblocks[j]) {
            reduce_unpack_block(b, jobs.part[j], jobs.nsplit[j], jobs.K[j], jobs.O[j], j | flags, Fin, H, M, dWg, dWc);
            return;
        }
        b -= jobs.nblocks[j];
    }
    EEG_DYN_SMEM(sm);
    reduce_bias_block(b, sm, jobs.bias_part, jobs.bias_B, H, jobs.dbg, jobs.dbc);
}

// Column sums of a dense (R x C) matrix in two fixed-order stages (projection bias gradient):
// stage 1: partial[chunk][C] over `rpc` rows per chunk (block = 2 row slices x 128 columns).
__global__ void colsum_partial_kernel(const float* __restrict__ A, int R, int C, int rpc, float* __restrict__ partial) {
    EEG_DYN_SMEM(sm);                                 // [2][128]
    const int c = threadIdx.x & 127, q = threadIdx.x >> 7;
    const int col = blockIdx.y * 128 + c;
    const int r0 = blockIdx.x * rpc, r1 = (r0 + rpc < R) ? r0 + rpc : R;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (col < C) {
        int r = r0 + q;
        for (; r + 6 < r1; r += 8) {
            a0 += A[(size_t)r * C + col];
            a1 += A[(size_t)(r + 2) * C + col];
            a2 += A[(size_t)(r + 4) * C + col];
            a3 += A[(size_t)(r + 6) * C + col];
        }
        for (; r < r1; r += 2) a0 += A[(size_t)r * C + col];
    }
    sm[q * 128 + c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q == 0 && col < C) partial[(size_t)blockIdx.x * C + col] = sm[c] + sm[128 + c];
}
// The same for C % 4 == 0, C <= 1024: a thread owns 4 consecutive columns (16-byte loads, 4 in flight) of one of
// 256 / (C/4) row slices; the slices are added in slice order through LDS.
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float* __restrict__ A, int R, int C, int rpc,
                                                              float* __restrict__ partial) {
    EEG_DYN_SMEM(sm);                                 // [nrs][C]
    const int c4n = C / 4, nrs = 256 / c4n;
    const int cq = threadIdx.x % c4n, q = threadIdx.x / c4n;
    const int r0 = blockIdx.x * rpc, r1 = (r0 + rpc < R) ? r0 + rpc : R;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (q < nrs) {
        const float* p = A + 4 * cq;
        int r = r0 + q;
        for (; r + 3 * nrs < r1; r += 4 * nrs) {
            a0 += ld4(p + (size_t)r * C);
            a1 += ld4(p + (size_t)(r + nrs) * C);
            a2 += ld4(p + (size_t)(r + 2 * nrs) * C);
            a3 += ld4(p + (size_t)(r + 3 * nrs) * C);
        }
        for (; r < r1; r += nrs) a0 += ld4(p + (size_t)r * C);
        st4(sm + q * C + 4 * cq, (a0 + a1) + (a2 + a3));
    }
    __syncthreads();
    for (int col = threadIdx.x; col < C; col += 256) {
        float s = 0.f;
        for (int i = 0; i < nrs; ++i) s += sm[i * C + col];
        partial[(size_t)blockIdx.x * C + col] = s;
    }
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int nchunk, int C, int split,
                                    float* __restrict__ out0, float* __restrict__ out1) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= C) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    for (; i + 4 <= nchunk; i += 4) {
        a0 += partial[(size_t)i * C + col];
        a1 += partial[(size_t)(i + 1) * C + col];
        a2 += partial[(size_t)(i + 2) * C + col];
        a3 += partial[(size_t)(i + 3) * C + col];
    }
    for (; i < nchunk; ++i) a0 += partial[(size_t)i * C + col];
    const float s = (a0 + a1) + (a2 + a3);
    if (col < split) out0[col] = s; else out1[col - split] = s;
}

// Hop-polynomial matrices: P_0 = I (implicit), then for every support S, in order:
//   b = S a; emit b; repeat K-1 times { c = 2 S b - a; emit c; (b, a) <- (c, b) };
// where `a` starts as I and is NOT reset between supports (reference quirk Q1, cell.py:83-93).
// supports: nsup pointers, each (Bs, N, N) with Bs = B (batched) or 1.  out: (Bs, M-1, N, N).
// One workgroup per graph; thread (i, j) owns one matrix element.
struct SupPtrs { const float* p[4]; };

__global__ void hop_polys_kernel(SupPtrs sup, int nsup, int N, int K, float* __restrict__ out) {
    EEG_DYN_SMEM(sm);
    float* S = sm;                       // [N*N]
    float* a = sm + kMaxNodes * kMaxNodes;
    float* b = a + kMaxNodes * kMaxNodes;
    float* c = b + kMaxNodes * kMaxNodes;
    const int g = blockIdx.x, tid = threadIdx.x;
    const int i = tid / N, j = tid % N;
    const bool on = tid < N * N;
    const int M1 = nsup * K;
    if (on) a[i * N + j] = (i == j) ? 1.f : 0.f;
    int emitted = 0;
    for (int s = 0; s < nsup; ++s) {
        __syncthreads();
        if (on) S[i * N + j] = sup.p[s][(size_t)g * N * N + i * N + j];
        __syncthreads();
        float v = 0.f;
        if (on) {
            for (int q = 0; q < N; ++q) v = fmaf(S[i * N + q], a[q * N + j], v);
            b[i * N + j] = v;
            out[((size_t)g * M1 + emitted) * N * N + i * N + j] = v;
        }
        ++emitted;
        for (int k = 2; k <= K; ++k) {
            __syncthreads();
            float w = 0.f;
            if (on) {
                for (int q = 0; q < N; ++q) w = fmaf(S[i * N + q], b[q * N + j], w);
                w = 2.f * w - a[i * N + j];
                c[i * N + j] = w;
                out[((size_t)g * M1 + emitted) * N * N + i * N + j] = w;
            }
            ++emitted;
            __syncthreads();
            if (on) {                       // (b, a) <- (c, b)
                const float nb = c[i * N + j], na = b[i * N + j];
                a[i * N + j] = na;
                b[i * N + j] = nb;
            }
        }
    }
}

}  // namespace eeg
