// Kernels of the spectral form of the hoisted x-part (see spec_common.h for the math and the layouts): basis (Jacobi), per-frequency
// weight packs, the two HBM-bound node mixes.  Included by spec_inst.cpp only.
#pragma once
#include "spec_common.h"

namespace eeg {

// info[0] = max |S - U diag(lam) U^T| (includes the asymmetry of S), info[1] = largest off-diagonal left after the sweeps,
// info[2] = max |S|.  LDS: 2 * 32 * 32 doubles + pair tables.
__global__ __launch_bounds__(256) void spectral_basis_kernel(const float* __restrict__ S, int N, float* __restrict__ out) {
    EEG_DYN_SMEM(sm);
    double* A = reinterpret_cast<double*>(sm);            // [32][32]
    double* V = A + 32 * 32;
    double* cs = V + 32 * 32;                             // [16][2]
    int* pq = reinterpret_cast<int*>(cs + 32);            // [16][2]
    double* red = reinterpret_cast<double*>(pq + 32);     // [256]
    const int tid = threadIdx.x;
    const int NP = (N + 1) & ~1, NH = NP / 2;
    for (int e = tid; e < 32 * 32; e += 256) {
        const int i = e >> 5, j = e & 31;
        A[e] = (i < N && j < N) ? 0.5 * ((double)S[i * N + j] + (double)S[j * N + i]) : 0.0;
        V[e] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int sweep = 0; sweep < kSpecSweeps; ++sweep) {
        for (int r = 0; r < NP - 1; ++r) {
            // round-robin tournament: pair 0 = (NP-1, r); pair k = ((r + k) mod (NP-1), (r - k) mod (NP-1))
            if (tid < NH) {
                int p = tid == 0 ? NP - 1 : (r + tid) % (NP - 1);
                int q = tid == 0 ? r : (r - tid + (NP - 1)) % (NP - 1);
                if (p > q) { const int t = p; p = q; q = t; }
                double c = 1.0, s = 0.0;
                if (q < N) {
                    const double apq = A[p * 32 + q];
                    if (fabs(apq) > 1e-300) {
                        const double theta = (A[q * 32 + q] - A[p * 32 + p]) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(t * t + 1.0);
                        s = t * c;
                    }
                }
                cs[2 * tid] = c; cs[2 * tid + 1] = s;
                pq[2 * tid] = p; pq[2 * tid + 1] = q;
            }
            __syncthreads();
            // columns: A <- A J, V <- V J   (the pairs are disjoint)
            for (int e = tid; e < NH * 32; e += 256) {
                const int k = e >> 5, i = e & 31, p = pq[2 * k], q = pq[2 * k + 1];
                if (q >= N || i >= N) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                const double ap = A[i * 32 + p], aq = A[i * 32 + q];
                A[i * 32 + p] = c * ap - s * aq;
                A[i * 32 + q] = s * ap + c * aq;
                const double vp = V[i * 32 + p], vq = V[i * 32 + q];
                V[i * 32 + p] = c * vp - s * vq;
                V[i * 32 + q] = s * vp + c * vq;
            }
            __syncthreads();
            // rows: A <- J^T A
            for (int e = tid; e < NH * 32; e += 256) {
                const int k = e >> 5, j = e & 31, p = pq[2 * k], q = pq[2 * k + 1];
                if (q >= N || j >= N) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                const double ap = A[p * 32 + j], aq = A[q * 32 + j];
                A[p * 32 + j] = c * ap - s * aq;
                A[q * 32 + j] = s * ap + c * aq;
            }
            __syncthreads();
        }
    }
    // outputs: U, T_m(lam), residual of the decomposition against the support AS GIVEN
    for (int e = tid; e < N * N; e += 256) out[e] = (float)V[(e / N) * 32 + (e % N)];
    for (int e = tid; e < kSpecTc; e += 256) {
        const int m = e >> 5, i = e & 31;
        double t = 0.0;
        if (i < N) {
            const double lam = A[i * 32 + i];
            double t0 = 1.0, t1 = lam;
            t = m == 0 ? t0 : t1;
            for (int k = 2; k <= m; ++k) { t = 2.0 * lam * t1 - t0; t0 = t1; t1 = t; }
        }
        out[N * N + e] = (float)t;
    }
    double rmax = 0.0, omax = 0.0, smax = 0.0;
    for (int e = tid; e < N * N; e += 256) {
        const int i = e / N, j = e % N;
        double rec = 0.0;
        for (int k = 0; k < N; ++k) rec += V[i * 32 + k] * A[k * 32 + k] * V[j * 32 + k];
        rmax = fmax(rmax, fabs((double)S[e] - rec));
        if (i != j) omax = fmax(omax, fabs(A[i * 32 + j]));
        smax = fmax(smax, fabs((double)S[e]));
    }
    for (int which = 0; which < 3; ++which) {
        __syncthreads();
        red[tid] = which == 0 ? rmax : (which == 1 ? omax : smax);
        __syncthreads();
        if (tid == 0) {
            double m = 0.0;
            for (int i = 0; i < 256; ++i) m = fmax(m, red[i]);
            out[N * N + kSpecTc + which] = (float)m;
        }
    }
    if (tid >= 3 && tid < kSpecInfo) out[N * N + kSpecTc + tid] = 0.f;
    // csum[i] = sum_n U[n][i]: U^T applied to a constant node vector -- a bias b (the same row for every node) is csum[i] * b in
    // the eigenbasis, which is how the grouped GEMM adds it
    if (tid < 32) {
        double c = 0.0;
        if (tid < N)
            for (int n = 0; n < N; ++n) c += V[n * 32 + tid];
        out[N * N + kSpecTc + kSpecInfo + tid] = (float)c;
    }
}
constexpr size_t kSpecBasisLds = (2 * 32 * 32 + 32 + 256) * sizeof(double) + 32 * sizeof(int);

__device__ __forceinline__ void pack_spectral_body(const float* __restrict__ Wg, const float* __restrict__ Wc, const float* __restrict__ basis,
                                                   float* __restrict__ out, const SpecPack& p, int bid, int nb) {
    const int Fin = p.Fin, H = p.H, M = p.M, N = p.N;
    const float* tc = basis + N * N;
    const NnqOrder ox = make_nnq_order(1, Fin), ot = make_nnq_order(1, 3 * H);
    const size_t stride = (size_t)nb * blockDim.x;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < p.total; idx += stride) {
        if (idx >= p.sxr) {                                    // row-major Wt_i[f][o], zero rows beyond Fin
            const size_t e0 = idx - p.sxr;
            const int i = (int)(e0 / p.sxr_stride), e = (int)(e0 - (size_t)i * p.sxr_stride), f = e / (3 * H), o = e - f * 3 * H;
            float v = 0.f;
            if (f < Fin)
                for (int m = 0; m < M; ++m) {
                    const float w = o < 2 * H ? Wg[((size_t)f * M + m) * (2 * H) + o] : Wc[((size_t)f * M + m) * H + (o - 2 * H)];
                    v = fmaf(tc[m * 32 + i], w, v);
                }
            out[idx] = v;
            continue;
        }
        const bool tr = idx >= p.sxtq;
        const size_t bs = tr ? p.sxtq_stride : p.sxq_stride, e0 = idx - (tr ? p.sxtq : p.sxq);
        const int i = (int)(e0 / bs);
        const size_t e = e0 - (size_t)i * bs;
        const int s4 = e & 3, lane = (e >> 2) & 63, nct = tr ? p.nct_t : p.nct_x;
        const int ct = (int)((e >> 8) % nct), c = (int)((e >> 8) / nct), j = 16 * ct + (lane & 15);
        const int k = nnq_k_of(tr ? ot : ox, c, lane >> 4, s4);
        float v = 0.f;
        if (k >= 0) {
            const int f = tr ? j : k, o = tr ? k : j;          // Wt_i[f][o]
            if (f < Fin) {
                for (int m = 0; m < M; ++m) {
                    const float w = o < 2 * H ? Wg[((size_t)f * M + m) * (2 * H) + o] : Wc[((size_t)f * M + m) * H + (o - 2 * H)];
                    v = fmaf(tc[m * 32 + i], w, v);
                }
            }
        }
        out[idx] = v;
    }
}
__global__ void pack_spectral_kernel(const float* __restrict__ Wg, const float* __restrict__ Wc, const float* __restrict__ basis,
                                     float* __restrict__ out, SpecPack p) {
    pack_spectral_body(Wg, Wc, basis, out, p, (int)blockIdx.x, (int)gridDim.x);
}

// ---- node mixes ------------------------------------------------------------------------------------------------------------------
// The node-major side is ALWAYS time-major: row r = t*B + b.  map = 1: the sample-major side is the BATCH-major model input
// (B, T, N, F) as the trainer holds it (model.py:253's transpose is never materialised): sample r lives at storage index b*T + t.
// Both mixes walk the node-major rows in order (N sequential streams); the sample-major side is touched in whole samples
// (N * F * 4 contiguous bytes each).
__device__ __forceinline__ size_t spec_sample(size_t r, int map, int T, int B) { return map ? (r % B) * (size_t)T + r / B : r; }

// in: sample-major X (S, N, F);  out: node-major Xh (N, Sp, F), Xh[i][r(s)] = sum_n U[n][i] X[s][n];  pad rows [S, Sp) <- 0
template <int N>
__global__ __launch_bounds__(256) void spec_mix_in_kernel(const float* __restrict__ X, const float* __restrict__ basis, int S, int Sp,
                                                         int F, int map, int T, int B, float* __restrict__ Xh) {
    const int F4 = F / 4, SPW = blockDim.x / F4;
    const int tl = threadIdx.x / F4, c4 = threadIdx.x % F4;
    if (tl >= SPW) return;
    const float* __restrict__ U = basis;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    float4* O4 = reinterpret_cast<float4*>(Xh);
    for (int r = blockIdx.x * SPW + tl; r < Sp; r += gridDim.x * SPW) {
        if (r >= S) {                                     // pad rows of every frequency
#pragma unroll
            for (int i = 0; i < N; ++i) O4[((size_t)i * Sp + r) * F4 + c4] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const size_t s = spec_sample((size_t)r, map, T, B);
        float4 x[N];
#pragma unroll
        for (int n = 0; n < N; ++n) x[n] = X4[(s * N + n) * F4 + c4];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const float u = U[n * N + i];
                a.x = fmaf(u, x[n].x, a.x);
                a.y = fmaf(u, x[n].y, a.y);
                a.z = fmaf(u, x[n].z, a.z);
                a.w = fmaf(u, x[n].w, a.w);
            }
            O4[((size_t)i * Sp + r) * F4 + c4] = a;
        }
    }
}

// in: node-major Yh (N, Sp, F);  out: sample-major Y (S, N, F), Y[s][j] = sum_i U[j][i] Yh[i][r(s)] (+ bias[F])
template <int N>
__global__ __launch_bounds__(256) void spec_mix_out_kernel(const float* __restrict__ Yh, const float* __restrict__ basis,
                                                          const float* __restrict__ bias, int S, int Sp, int F, int map, int T, int B,
                                                          float* __restrict__ Y) {
    const int F4 = F / 4, SPW = blockDim.x / F4;
    const int tl = threadIdx.x / F4, c4 = threadIdx.x % F4;
    if (tl >= SPW) return;
    const float* __restrict__ U = basis;
    const float4* I4 = reinterpret_cast<const float4*>(Yh);
    float4* O4 = reinterpret_cast<float4*>(Y);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias != nullptr) bv = reinterpret_cast<const float4*>(bias)[c4];
    for (int r = blockIdx.x * SPW + tl; r < S; r += gridDim.x * SPW) {
        const size_t s = spec_sample((size_t)r, map, T, B);
        float4 y[N];
#pragma unroll
        for (int i = 0; i < N; ++i) y[i] = I4[((size_t)i * Sp + r) * F4 + c4];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float4 a = bv;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const float u = U[j * N + i];
                a.x = fmaf(u, y[i].x, a.x);
                a.y = fmaf(u, y[i].y, a.y);
                a.z = fmaf(u, y[i].z, a.z);
                a.w = fmaf(u, y[i].w, a.w);
            }
            O4[(s * N + j) * F4 + c4] = a;
        }
    }
}

// Both mixes on the matrix pipe (round 6): per unit (row r, 32-column tile) the mix is a (N x N) x (N x 32) product -- A = the
// basis (zero-padded to 32 x 2*KS, in registers), B = 2*KS rows of 32 consecutive floats (one 128-byte line per half-wave and
// k-step, straight from HBM into registers, no LDS), D = 32 x 32 of which the first N rows are stored (128-byte lines again).
// The VALU form above spends 4 N^2 FMAs per 16-byte column (18 us of VALU per 150 MB pass at N = 19, scalar loads of the 361
// coefficients in the loop); here a unit is KS MFMAs.  DIR 0: to nodes (X -> Xh = U^T X, pad rows written as zeros); DIR 1: from
// nodes (Yh -> Y = U Yh + bias).  Row maps as above.
template <int DIR, int KS>
__global__ __launch_bounds__(256) void spec_mix_mfma_kernel(const float* __restrict__ in, const float* __restrict__ basis,
                                                           const float* __restrict__ bias, int N, int S, int Sp, int F, int map, int T,
                                                           int B, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, hh = lane >> 5, l32 = lane & 31;
    const int gw = (int)blockIdx.x * 4 + wave_uniform((int)threadIdx.x >> 6), nw = (int)gridDim.x * 4;
    // A[o][k] (o = output node / frequency = l32, k = 2 ks + hh): DIR 0: U[k][o]; DIR 1: U[o][k]
    float a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 2 * ks + hh;
        a[ks] = (k < N && l32 < N) ? (DIR == 0 ? basis[k * N + l32] : basis[l32 * N + k]) : 0.f;
    }
    const int NFT = ceil_div(F, 32), rows = DIR == 0 ? Sp : S, units = rows * NFT;
    for (int u = gw; u < units; u += nw) {
        const int r = u / NFT, ft = u - r * NFT, f = 32 * ft + l32, fc = f < F ? f : F - 1;
        const size_t s = r < S ? spec_sample((size_t)r, map, T, B) : 0;
        f32x16 acc;
        const float bv = (DIR == 1 && bias != nullptr) ? bias[fc] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = bv;
        if (r < S) {
            float b[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                int k = 2 * ks + hh;
                if (k >= N) k = N - 1;                    // (its A column is zero: any finite in-bounds value)
                b[ks] = DIR == 0 ? in[(s * N + k) * F + fc] : in[((size_t)k * Sp + r) * F + fc];
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = mfma32(a[ks], b[ks], acc);
        }
        if (f < F) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int o = 8 * (v >> 2) + 4 * hh + (v & 3);
                if (2 * KS <= 8 * (v >> 2)) continue;     // (output rows beyond the padded node count: never any)
                if (o < N) {
                    if (DIR == 0) out[((size_t)o * Sp + r) * F + f] = acc[v];
                    else out[(s * N + o) * F + f] = acc[v];
                }
            }
        }
    }
}

// rows [S, Sp) of every frequency of a node-major (N, Sp, F) tensor <- 0 (a producer that writes the S real rows only)
__global__ void spec_zero_pad_kernel(float* __restrict__ Xh, int N, int S, int Sp, int F) {
    const int per = (Sp - S) * F, total = N * per;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int i = e / per, r = e - i * per;
        Xh[((size_t)i * Sp + S) * F + r] = 0.f;
    }
}

// The same two mixes for any node count (rolled loops, coefficients from LDS): montages other than the 19-electrode one.
__global__ __launch_bounds__(256) void spec_mix_generic_kernel(const float* __restrict__ in, const float* __restrict__ basis,
                                                              const float* __restrict__ bias, int N, int S, int Sp, int F, int map,
                                                              int T, int B, int to_nodes, float* __restrict__ out) {
    EEG_DYN_SMEM(sm);                                     // U [N*N]
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) sm[e] = basis[e];
    __syncthreads();
    const int F4 = F / 4;
    const size_t total = (size_t)(to_nodes ? Sp : S) * N * F4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % F4), o = (int)((e / F4) % N);
        const size_t r = e / ((size_t)F4 * N);
        const size_t s = r < (size_t)S ? spec_sample(r, map, T, B) : r;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (to_nodes) {                                   // out = Xh[o][r] = sum_n U[n][o] X[s(r)][n]
            if (r < (size_t)S) {
                for (int n = 0; n < N; ++n) {
                    const float u = sm[n * N + o];
                    const float4 x = reinterpret_cast<const float4*>(in)[(s * N + n) * F4 + c4];
                    a.x = fmaf(u, x.x, a.x); a.y = fmaf(u, x.y, a.y); a.z = fmaf(u, x.z, a.z); a.w = fmaf(u, x.w, a.w);
                }
            }
            reinterpret_cast<float4*>(out)[((size_t)o * Sp + r) * F4 + c4] = a;
        } else {                                          // out = Y[s(r)][o] = sum_i U[o][i] Yh[i][r] + bias
            if (bias != nullptr) a = reinterpret_cast<const float4*>(bias)[c4];
            for (int i = 0; i < N; ++i) {
                const float u = sm[o * N + i];
                const float4 y = reinterpret_cast<const float4*>(in)[((size_t)i * Sp + r) * F4 + c4];
                a.x = fmaf(u, y.x, a.x); a.y = fmaf(u, y.y, a.y); a.z = fmaf(u, y.z, a.z); a.w = fmaf(u, y.w, a.w);
            }
            reinterpret_cast<float4*>(out)[(s * N + o) * F4 + c4] = a;
        }
    }
}

}  // namespace eeg
