// Graph diffusion (node mixing) on the 19-electrode graph.
//
//  * lds_load_polys / lds_diffuse_tiles: building blocks shared with the recurrent kernels —
//    the (M-1) non-identity hop-polynomial matrices of one graph live in LDS zero-padded to
//    32x32 and are applied to an LDS-resident (32 x W) feature tile with fp32 MFMA
//    (D[n][f] = sum_n' P_m[n][n'] X[n'][f], or P_m^T for the adjoint).
//  * diffuse_fwd_kernel: the hoisted, HBM-bound diffusion step over all T*B samples:
//    reads X (S,N,F) once, writes the M-1 hop planes (M-1,S,N,F).  Algorithmic bytes per
//    sample = 4*N*F*M.
//  * diffuse_adj_kernel: dX[s] = Z_0[s] + sum_{m>=1} P_m^T Z_m[s] for Z (S,N,M*F).
#pragma once
#include "common.h"
#include "lds_diffuse.h"

namespace eeg {

// ---- standalone forward diffusion: X (S,N,F) -> planes (M-1,S,N,F) --------------------------
// grid = (G, TG): workgroup (g, tg) owns graph g (g = clip b when P is per-clip, else a slice of
// the sample range) so its polynomials are staged in LDS once; it walks its samples with the next
// sample's rows prefetched into registers while the current one is diffused (HBM latency hidden
// inside the workgroup, 3-4 workgroups per CU hide the rest).
// LDS: Pl | tile [NR][FS] with NR = round_up(N,4), FS = lds_stride(M * FP), FP = round_up(F,16):
// slot 0 = X, slots 1..M-1 = results (staged so the global stores are full coalesced rows).
constexpr int kDiffPrefetch = 4;        // float4 per thread per sample: covers N*F <= 4096 floats

// Wide rows are processed in column chunks: this launch covers columns [c0, c0+Fc) of the F-wide rows.
__global__ __launch_bounds__(256) void diffuse_fwd_kernel(const float* __restrict__ X, const float* __restrict__ P,
                                                          int p_batched, int S, int B, int N, int F, int M,
                                                          float* __restrict__ planes, size_t plane_stride, int c0, int Fc) {
    EEG_DYN_SMEM(sm);
    const int FP = round_up(Fc, 16), FS = lds_stride(M * FP);
    float* Pl = sm;
    float* tile = sm + (M - 1) * kPFloats;
    const int tid = threadIdx.x, NR = round_up(N, 4);
    const int nf4 = Fc / 4, nq = N * nf4;           // Fc % 4 == 0, nq <= 1024 (the host sizes the chunk)
    // sample walk: per-clip graphs -> s = t*B + g for t = tg, tg+TG, ...; shared graph -> s = wg, wg+nwg, ...
    int s_first, s_step, g;
    if (p_batched) {
        g = blockIdx.x;
        s_first = blockIdx.y * B + g;
        s_step = gridDim.y * B;
    } else {
        g = 0;
        s_first = blockIdx.y * gridDim.x + blockIdx.x;
        s_step = gridDim.x * gridDim.y;
    }
    for (int e = tid; e < NR * FS; e += 256) tile[e] = 0.f;
    lds_load_polys(Pl, P, g, M, N);

    float4 pre[kDiffPrefetch];
    auto fetch = [&](int s) {
        const float* src = X + (size_t)s * N * F + c0;
#pragma unroll
        for (int i = 0; i < kDiffPrefetch; ++i) {
            const int q = tid + 256 * i;
            pre[i] = (s < S && q < nq) ? *reinterpret_cast<const float4*>(src + (size_t)(q / nf4) * F + 4 * (q % nf4))
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(s_first);
    for (int s = s_first; s < S; s += s_step) {
        __syncthreads();                              // tile free: previous stores have read it
#pragma unroll
        for (int i = 0; i < kDiffPrefetch; ++i) {
            const int q = tid + 256 * i;
            if (q < nq) {
                float* d = tile + (q / nf4) * FS + 4 * (q % nf4);
                d[0] = pre[i].x; d[1] = pre[i].y; d[2] = pre[i].z; d[3] = pre[i].w;
            }
        }
        fetch(s + s_step);                            // next sample's rows fly during the MFMAs
        __syncthreads();
        lds_diffuse_tiles<false>(tile, FS, 0, FP, FP, FP, Pl, M, N, NR);
        __syncthreads();
        for (int m1 = 0; m1 < M - 1; ++m1) {
            float* dst = planes + (size_t)m1 * plane_stride + (size_t)s * N * F + c0;
            for (int q = tid; q < nq; q += 256) {
                const float* t = tile + (q / nf4) * FS + FP * (m1 + 1) + 4 * (q % nf4);
                *reinterpret_cast<float4*>(dst + (size_t)(q / nf4) * F + 4 * (q % nf4)) = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
    }
}

// ---- streaming forward diffusion (the HBM-roofline kernel) ------------------------------------
// One thread = one 16-byte feature column (4 features) of one sample: it loads the N node rows of
// that column into registers (N independent 16-byte loads in flight per thread), applies the
// (M-1) hop polynomials with VALU FMAs and streams the results out.  All threads of a workgroup
// work on samples of ONE graph (g = blockIdx.x), so the polynomial coefficients are wave-uniform:
// the compiler keeps them in SGPRs (scalar loads), there is no LDS and no barrier.
// Algorithmic bytes per sample: 4*N*F*M; VALU work 2*(M-1)*N*N*F flop is ~4x below the HBM time.
// x_bt != 0: X is batch-major (B, S/B, N, F) (the model input as the trainer holds it).  x_bt == 1: the planes
// are written time-major and xcopy (optional) receives the time-major copy of X as a by-product; x_bt == 2: the
// planes keep X's batch-major order (every workgroup then reads AND writes one contiguous stretch per clip; the
// GEMMs that consume X and the planes address their rows through a (b,t) map) -- exactly 4*N*F*M bytes per sample.
template <int N>
__global__ __launch_bounds__(256) void diffuse_fwd_stream_kernel(const float* __restrict__ X,
                                                                 const float* __restrict__ P, int p_batched,
                                                                 int S, int B, int F, int M,
                                                                 float* __restrict__ planes, size_t plane_stride,
                                                                 int x_bt, float* __restrict__ xcopy) {
    const int F4 = F / 4, SPW = blockDim.x / F4;     // float4 columns per sample, samples per pass
    const int tl = threadIdx.x / F4, c4 = threadIdx.x % F4;
    const int sB = p_batched ? B : 1, g = p_batched ? blockIdx.x : 0;
    const int T = S / sB, Tc = S / B;                // samples of this graph; time steps per clip
    const float* __restrict__ Pg = P + (size_t)g * (M - 1) * N * N;
    if (tl >= SPW) return;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    float4* O4 = reinterpret_cast<float4*>(planes);
    float4* C4 = reinterpret_cast<float4*>(xcopy);
    for (int t = blockIdx.y * SPW + tl; t < T; t += gridDim.y * SPW) {
        const size_t s = (size_t)t * sB + g;         // time-major sample index (t_clip * B + b)
        const size_t ssrc = x_bt ? (s % B) * Tc + s / B : s;
        const size_t sdst = x_bt == 2 ? ssrc : s;   // x_bt == 2: the planes keep the batch-major sample order of X
        float4 x[N];
#pragma unroll
        for (int n = 0; n < N; ++n) x[n] = X4[(ssrc * N + n) * F4 + c4];
        if (xcopy != nullptr) {
#pragma unroll
            for (int n = 0; n < N; ++n) C4[(s * N + n) * F4 + c4] = x[n];
        }
        for (int m1 = 0; m1 < M - 1; ++m1) {
            const float* __restrict__ Pm = Pg + m1 * N * N;
            float4* out = O4 + (size_t)m1 * (plane_stride / 4) + (sdst * N) * F4 + c4;
#pragma unroll
            for (int n = 0; n < N; ++n) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < N; ++q) {
                    const float p = Pm[n * N + q];
                    a.x = fmaf(p, x[q].x, a.x);
                    a.y = fmaf(p, x[q].y, a.y);
                    a.z = fmaf(p, x[q].z, a.z);
                    a.w = fmaf(p, x[q].w, a.w);
                }
                out[(size_t)n * F4] = a;
            }
        }
    }
}

// ---- streaming adjoint (same idea as diffuse_fwd_stream_kernel) -------------------------------
// dX[s][n] = Z_0[s][n] + sum_{m>=1} sum_q P_m[q][n] Z_m[s][q]  [+ add[s][n]],  Z (S,N,M*F).
// One thread = one 16-byte feature column of one sample; the hop planes are consumed one at a time
// (N loads in flight, N accumulators), coefficients are wave-uniform scalars.  Algorithmic bytes per
// sample: 4*N*F*(M+1).
template <int N>
__global__ __launch_bounds__(256, 3) void diffuse_adj_stream_kernel(const float* __restrict__ Z,
                                                                    const float* __restrict__ P, int p_batched,
                                                                    int S, int B, int F, int M,
                                                                    const float* __restrict__ add,
                                                                    float* __restrict__ dX) {
    const int F4 = F / 4, SPW = blockDim.x / F4;
    const int tl = threadIdx.x / F4, c4 = threadIdx.x % F4;
    const int sB = p_batched ? B : 1, g = p_batched ? blockIdx.x : 0;
    const int T = S / sB;
    const float* __restrict__ Pg = P + (size_t)g * (M - 1) * N * N;
    if (tl >= SPW) return;
    const float4* Z4 = reinterpret_cast<const float4*>(Z);
    const float4* A4 = reinterpret_cast<const float4*>(add);
    float4* D4 = reinterpret_cast<float4*>(dX);
    const unsigned zrow = (unsigned)(M * F4);               // float4 per node row of Z
    constexpr int NCH = 4, NA = (N + NCH - 1) / NCH;        // operands are consumed in NCH chunks of NA nodes:
    for (int t = blockIdx.y * SPW + tl; t < T; t += gridDim.y * SPW) {   // N accumulators + NA operands in flight
        const unsigned s = (unsigned)t * sB + g;
        // per-lane 32-bit element offsets; every load below is (wave-uniform pointer)[lane offset]
        const unsigned zl = s * N * zrow + c4, xl = s * N * F4 + c4;        // < 2^32 float4 (host-checked)
        float4 acc[N];
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = (Z4 + n * zrow)[zl];
        EEG_SCHED_FENCE();
        if (add != nullptr) {
#pragma unroll
            for (int h = 0; h < NCH; ++h) {
                float4 a[NA];
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    if (h * NA + q < N) a[q] = (A4 + (h * NA + q) * F4)[xl];
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    if (h * NA + q < N) {
                        float4& d = acc[h * NA + q];
                        d.x += a[q].x; d.y += a[q].y; d.z += a[q].z; d.w += a[q].w;
                    }
                EEG_SCHED_FENCE();
            }
        }
        for (int m1 = 0; m1 < M - 1; ++m1) {
            const float* __restrict__ Pm = Pg + m1 * N * N;
            const float4* zp = Z4 + (m1 + 1) * F4;
#pragma unroll
            for (int h = 0; h < NCH; ++h) {
                const int q0 = h * NA, nq = (N - q0) < NA ? (N - q0) : NA;
                float4 z[NA];
#pragma unroll
                for (int q = 0; q < NA; ++q)
                    if (q < nq) z[q] = (zp + (q0 + q) * zrow)[zl];
#pragma unroll
                for (int q = 0; q < NA; ++q) {
                    if (q >= nq) continue;
#pragma unroll
                    for (int n = 0; n < N; ++n) {
                        const float p = Pm[(q0 + q) * N + n];   // P_m^T[n][q]
                        acc[n].x = fmaf(p, z[q].x, acc[n].x);
                        acc[n].y = fmaf(p, z[q].y, acc[n].y);
                        acc[n].z = fmaf(p, z[q].z, acc[n].z);
                        acc[n].w = fmaf(p, z[q].w, acc[n].w);
                    }
                }
                EEG_SCHED_FENCE();
            }
        }
#pragma unroll
        for (int n = 0; n < N; ++n) (D4 + n * F4)[xl] = acc[n];
    }
}

// ---- the same adjoint, walking Z in STORAGE ORDER (round 5) ------------------------------------------------------------------
// A Z row (node q of sample s) holds the MM hop slots side by side (MM * F floats).  The kernel above consumes one hop plane at a
// time, i.e. every load instruction of a wave takes F floats out of every MM * F (a third of a row at MM = 3) and returns to the same
// DRAM pages MM times.  Here a thread owns the same 16-byte column but walks the rows of its sample once, top to bottom: the MM
// slot pieces of node q are requested together, D nodes ahead of their use, and scattered into the N accumulators with the
// wave-uniform coefficients P_m[q][n] (row q of P_m = column q of P_m^T).  Same arithmetic per output element, other sum order
// inside an element only (hop-major -> node-major; fp32, ~1e-7 relative).
template <int N, int MM>
__global__ __launch_bounds__(256, 2) void diffuse_adj_rows_kernel(const float* __restrict__ Z,
                                                                  const float* __restrict__ P, int p_batched,
                                                                  int S, int B, int F,
                                                                  const float* __restrict__ add,
                                                                  float* __restrict__ dX) {
    constexpr int D = MM <= 3 ? 3 : 2;                      // node rows in flight ahead of the one being consumed (registers)
    const int F4 = F / 4, SPW = blockDim.x / F4;
    const int tl = threadIdx.x / F4, c4 = threadIdx.x % F4;
    const int sB = p_batched ? B : 1, g = p_batched ? blockIdx.x : 0;
    const int T = S / sB;
    const float* __restrict__ Pg = P + (size_t)g * (MM - 1) * N * N;
    if (tl >= SPW) return;
    const float4* Z4 = reinterpret_cast<const float4*>(Z);
    const float4* A4 = reinterpret_cast<const float4*>(add);
    float4* D4 = reinterpret_cast<float4*>(dX);
    const unsigned zrow = (unsigned)(MM * F4);
    for (int t = blockIdx.y * SPW + tl; t < T; t += gridDim.y * SPW) {
        const unsigned s = (unsigned)t * sB + g;
        const unsigned zl = s * N * zrow + c4, xl = s * N * F4 + c4;        // < 2^32 float4 (host-checked)
        float4 acc[N];
        if (add != nullptr) {
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = (A4 + n * F4)[xl];
        } else {
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // D node rows in flight; the walk over the rows is a ROLLED loop in groups of D (fully unrolled, the compiler interleaves the
        // multiply-adds of many rows and spills 200 registers), so the identity hop is a one-hot coefficient row like the others
        float4 zb[D][MM];
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int m = 0; m < MM; ++m) zb[j][m] = (Z4 + j * zrow + m * F4)[zl];
        auto consume = [&](int q, const float4 (&z)[MM]) __attribute__((always_inline)) {
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const float e = n == q ? 1.f : 0.f;                          // hop 0 = the identity (wave-uniform select)
                acc[n].x = fmaf(e, z[0].x, acc[n].x);
                acc[n].y = fmaf(e, z[0].y, acc[n].y);
                acc[n].z = fmaf(e, z[0].z, acc[n].z);
                acc[n].w = fmaf(e, z[0].w, acc[n].w);
            }
#pragma unroll
            for (int m1 = 0; m1 < MM - 1; ++m1) {
                const float* __restrict__ Pm = Pg + m1 * N * N + q * N;      // row q of P_m
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    const float p = Pm[n];
                    acc[n].x = fmaf(p, z[m1 + 1].x, acc[n].x);
                    acc[n].y = fmaf(p, z[m1 + 1].y, acc[n].y);
                    acc[n].z = fmaf(p, z[m1 + 1].z, acc[n].z);
                    acc[n].w = fmaf(p, z[m1 + 1].w, acc[n].w);
                }
            }
        };
        constexpr int NG = N / D;                           // whole groups; the N % D rows behind them are already in flight
#pragma unroll 1
        for (int q0 = 0; q0 < NG * D; q0 += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                float4 z[MM];
#pragma unroll
                for (int m = 0; m < MM; ++m) z[m] = zb[j][m];
                const int qn = q0 + j + D < N ? q0 + j + D : N - 1;         // (past the end: a valid row, never consumed)
#pragma unroll
                for (int m = 0; m < MM; ++m) zb[j][m] = (Z4 + qn * zrow + m * F4)[zl];
                EEG_SCHED_FENCE();
                consume(q0 + j, z);
                EEG_SCHED_FENCE();
            }
        }
#pragma unroll
        for (int j = 0; j < N - NG * D; ++j) consume(NG * D + j, zb[j]);
#pragma unroll
        for (int n = 0; n < N; ++n) (D4 + n * F4)[xl] = acc[n];
    }
}

// ---- standalone adjoint: Z (S,N,M*F) -> dX (S,N,F) = Z_0 + sum_m P_m^T Z_m [+ add] -----------
// LDS tile [NR][ZS], NR = round_up(N,4), ZS = lds_stride(M*FP): slot m holds Z_m (cols padded to FP).
// Column chunk [c0, c0+Fc) of every F-wide slot per launch (wide rows).
__global__ __launch_bounds__(256) void diffuse_adj_kernel(const float* __restrict__ Z, const float* __restrict__ P,
                                                          int p_batched, int S, int B, int N, int F, int M,
                                                          const float* __restrict__ add, float* __restrict__ dX, int c0, int Fc) {
    EEG_DYN_SMEM(sm);
    const int FP = round_up(Fc, 16), ZS = lds_stride(M * FP);
    float* Pl = sm;
    float* tile = sm + (M - 1) * kPFloats;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int lr = lane & 15, lg = lane >> 4, NR = round_up(N, 4);
    for (int e = tid; e < NR * ZS; e += blockDim.x) tile[e] = 0.f;
    if (!p_batched) lds_load_polys(Pl, P, 0, M, N);
    const int nf4 = Fc / 4, nct = FP / 16, nks = ceil_div(N, 4);
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        __syncthreads();
        if (p_batched) lds_load_polys(Pl, P, s % B, M, N);
        const float* src = Z + (size_t)s * N * M * F + c0;
        for (int q = tid; q < N * M * nf4; q += blockDim.x) {
            const int n = q / (M * nf4), rem = q % (M * nf4), m = rem / nf4, c4 = rem % nf4;
            const float4 v = *reinterpret_cast<const float4*>(src + ((size_t)n * M + m) * F + 4 * c4);
            float* d = tile + n * ZS + m * FP + 4 * c4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        // out tile (rt, ct): acc = Z_0 tile; acc += P_m^T Z_m over m >= 1
        for (int t = wave; t < 2 * nct; t += nwaves) {
            const int ct = t % nct, rt = t / nct;
            f32x4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = rt * 16 + 4 * lg + r;
                acc[r] = n < N ? tile[n * ZS + ct * 16 + lr] : 0.f;
            }
            for (int m1 = 0; m1 < M - 1; ++m1) {
                const float* Pm = Pl + m1 * kPFloats;
                for (int ks = 0; ks < nks; ++ks) {
                    const int kk = 4 * ks + lg;
                    acc = mfma16(Pm[kk * kPStride + rt * 16 + lr], tile[kk * ZS + (m1 + 1) * FP + ct * 16 + lr], acc);
                }
            }
            const int col = ct * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = rt * 16 + 4 * lg + r;
                if (n < N && col < Fc) {
                    const size_t o = ((size_t)s * N + n) * F + c0 + col;
                    dX[o] = add != nullptr ? acc[r] + add[o] : acc[r];
                }
            }
        }
    }
}

}  // namespace eeg
