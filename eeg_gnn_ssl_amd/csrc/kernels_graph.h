// Per-clip correlation graph -> dual random-walk supports, on the GPU.
//
// Reference (CPU, inside the DataLoader workers): dataloader_detection.py:258-307 (_get_indiv_graphs:
// normalised 'valid' cross-correlation of every electrode pair of the (N, T*D) clip = |cosine Gram|,
// diagonal 1), data_utils.py:174-200 (keep_topk, top_k = 3, directed), utils.py:220-230 +
// dataloader_detection.py:346-349 (S1 = (D^-1 A)^T, S2 = (D_in^-1 A^T)^T).
//
// This is the memory-bound variant of the path: each 456 kB clip is read once for ~2.3 MFLOP.
//  * corr_gram_kernel: grid (B, NS).  Workgroup (b, sp) accumulates the Gram of clip b over its
//    share of the time steps with fp32 MFMA straight from global memory: lane (i, g) loads the 16
//    bytes X[t][node i][16q+4g .. +3] (rows of 400 B are consumed in 64-byte pieces by the 4 lane
//    groups) and feeds them as BOTH operands (G = X X^T), one float per MFMA; the K order inside a
//    16-feature chunk is permuted, which a sum over all k does not care about.  No LDS staging.
//    Algorithmic bytes: 4*T*N*D per clip.
//  * corr_finish_kernel: one workgroup per clip: fixed-order sum of the NS partial Grams,
//    normalisation, |.|, diag = 1, top-k per row, S1/S2.
#pragma once
#include "common.h"

namespace eeg {

constexpr int kGramTile = 256;                 // one 16x16 MFMA accumulator tile, C layout (r*64 + lane)
constexpr int kGramFloats = 3 * kGramTile;     // tiles (0,0), (0,1), (1,1) of the padded 32x32 Gram

// NQ = number of 16-feature chunks (compile-time so that all loads of a time step are issued
// together and the next step's loads fly during the MFMAs of the current one).
template <int NQ>
__global__ __launch_bounds__(256) void corr_gram_kernel(const float* __restrict__ X, int T, int N, int D,
                                                        float* __restrict__ part) {
    EEG_DYN_SMEM(sm);                          // [4 waves][kGramFloats]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int b = blockIdx.x, sp = blockIdx.y, NS = gridDim.y;
    f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = {0.f, 0.f, 0.f, 0.f}, c11 = {0.f, 0.f, 0.f, 0.f};
    const bool has0 = i < N, has1 = 16 + i < N;
    float4 cur0[NQ], cur1[NQ], nxt0[NQ], nxt1[NQ];
    auto fetch = [&](int t, float4* a0, float4* a1) {
        const float* xt = X + ((size_t)b * T + t) * N * D;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int f = 16 * q + 4 * g;          // D % 4 == 0 (checked by the host)
            const bool ok = t < T && f < D;
            a0[q] = (ok && has0) ? *reinterpret_cast<const float4*>(xt + (size_t)i * D + f) : make_float4(0.f, 0.f, 0.f, 0.f);
            a1[q] = (ok && has1) ? *reinterpret_cast<const float4*>(xt + (size_t)(16 + i) * D + f) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int t0 = sp * 4 + wave, dt = 4 * NS;
    fetch(t0, cur0, cur1);
    for (int t = t0; t < T; t += dt) {
        fetch(t + dt, nxt0, nxt1);
        EEG_SCHED_FENCE();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 a0 = cur0[q], a1 = cur1[q];
            c00 = mfma16(a0.x, a0.x, c00); c01 = mfma16(a0.x, a1.x, c01); c11 = mfma16(a1.x, a1.x, c11);
            c00 = mfma16(a0.y, a0.y, c00); c01 = mfma16(a0.y, a1.y, c01); c11 = mfma16(a1.y, a1.y, c11);
            c00 = mfma16(a0.z, a0.z, c00); c01 = mfma16(a0.z, a1.z, c01); c11 = mfma16(a1.z, a1.z, c11);
            c00 = mfma16(a0.w, a0.w, c00); c01 = mfma16(a0.w, a1.w, c01); c11 = mfma16(a1.w, a1.w, c11);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) { cur0[q] = nxt0[q]; cur1[q] = nxt1[q]; }
    }
    float* mine = sm + wave * kGramFloats;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        mine[0 * kGramTile + r * 64 + lane] = c00[r];
        mine[1 * kGramTile + r * 64 + lane] = c01[r];
        mine[2 * kGramTile + r * 64 + lane] = c11[r];
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
