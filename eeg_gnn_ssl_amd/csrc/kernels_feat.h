// Input featurisation on the GPU: 1-second windows -> log|FFT| of the positive
// frequencies -> optional augmentation (left/right reflection = node permutation, amplitude jitter =
// additive log scale) -> z-score.  Reference (CPU, in the DataLoader): data_utils.py:13-35 computeFFT
// (scipy.fftpack, float64), dataloader_detection.py:57-71 (windowing), :233-256 (augmentation),
// utils.py:393-428 (StandardScaler).
//
// The reference transforms float64 signals; log amplitudes of weak bins amplify any error of the
// transform, so the DFT here runs in fp64 (the MI355X vector unit does 39 T fp64 FMA/s; the whole
// kernel is ~0.5 ms for 256 one-minute clips).  One wave per (clip, node, window group): the W samples
// of a window sit in LDS as doubles (broadcast reads), lane k < W/4+1 owns the frequency pair
// (k, W/2-k), whose twiddles differ only by (-1)^n: four sums over even / odd samples give both bins.
// Twiddles advance by complex rotation (error ~W*eps).  Needs W % 4 == 0, W/4 + 1 <= 64.
#pragma once
#include "common.h"

namespace eeg {

// source channel of node nd under the reflection augmentation; an entry outside 0..N-1 (not a permutation: caller error) falls back to
// the node itself instead of addressing another clip's signals / features
__device__ __forceinline__ int perm_source(const int* __restrict__ perm, int b, int N, int nd) {
    if (perm == nullptr) return nd;
    const int s = perm[b * N + nd];
    return (unsigned)s < (unsigned)N ? s : nd;
}

__global__ __launch_bounds__(64) void fft_features_kernel(const float* __restrict__ raw, int N, int T, int W, int tchunk,
                                                          const int* __restrict__ perm, const float* __restrict__ log_scale,
                                                          float mean, float inv_std, float* __restrict__ feat_raw,
                                                          float* __restrict__ feat_std) {
    EEG_DYN_SMEM(sm);
    double* xs = reinterpret_cast<double*>(sm);          // [W]
    const int lane = threadIdx.x, b = blockIdx.x / N, nd = blockIdx.x % N, H2 = W / 2;
    const int src = perm_source(perm, b, N, nd);            // EEG_seq_reflect[:, pair] = EEG_seq[:, swapped pair]
    const double ls = log_scale != nullptr ? (double)log_scale[b] : 0.0;
    const float* sig = raw + ((size_t)b * N + src) * (size_t)T * W;
    const bool active = lane <= W / 4;
    double c1 = 1.0, s1 = 0.0;
    if (active) sincos(6.283185307179586476925286766559 * (double)lane / (double)W, &s1, &c1);
    const int t0 = blockIdx.y * tchunk, t1 = (t0 + tchunk < T) ? t0 + tchunk : T;
    for (int t = t0; t < t1; ++t) {
        EEG_WAVE_SYNC();                                  // previous window fully consumed
        for (int i = lane; i < W; i += 64) xs[i] = (double)sig[(size_t)t * W + i];
        EEG_WAVE_SYNC();
        if (active) {
            double ec = 0.0, es = 0.0, oc = 0.0, os = 0.0, c = 1.0, s = 0.0;
            for (int n = 0; n < W; n += 2) {
                const double x0 = xs[n], x1 = xs[n + 1];
                ec = fma(x0, c, ec); es = fma(x0, s, es);
                double cn = c * c1 - s * s1, sn = s * c1 + c * s1;
                oc = fma(x1, cn, oc); os = fma(x1, sn, os);
                c = cn * c1 - sn * s1; s = sn * c1 + cn * s1;
            }
            // the un-augmented features stay at the source channel (perm is a permutation: every slot is written once)
            const size_t o_raw = (((size_t)b * T + t) * N + src) * H2, o_std = (((size_t)b * T + t) * N + nd) * H2;
            auto emit = [&](int k, double re, double im) {
                double amp = sqrt(re * re + im * im);
                if (amp == 0.0) amp = 1e-8;                // computeFFT: avoid log of 0
                const double v = log(amp);
                if (feat_raw != nullptr) feat_raw[o_raw + k] = (float)v;
                if (feat_std != nullptr) feat_std[o_std + k] = (float)(((v + ls) - (double)mean) * (double)inv_std);
            };
            if (lane < H2) emit(lane, ec + oc, es + os);
            if (lane > 0 && H2 - lane > lane && H2 - lane < H2) emit(H2 - lane, ec - oc, es - os);
        }
    }
}


// ---- W = 200 (the reference's 200 Hz x 1-s steps, dataloader_detection.py:54-67): mixed-radix real transform -----------------
// The direct DFT above does 2 x 200 x 100 multiply-adds per window (23 GFLOP fp64 for 256 one-minute clips = 5 x the HBM floor
// of this kernel).  Here: z[n] = x[2n] + i x[2n+1] (n < 100), Z = DFT_100(z) as 10 x 10 Cooley-Tukey with every 10-point DFT done
// in registers as a 2 x 5 prime-factor transform (no twiddles inside), then the real-input split
//     2 X[k] = (Z[k] + conj Z[100-k]) - i e^{-2 pi i k / 200} (Z[k] - conj Z[100-k]),   k = 0..99,
// all in fp64 (the log of a weak bin amplifies any error of the transform, see above): ~3 k fp64 operations per window instead of
// 160 k.  Ten lanes own one window, six windows per wave (lanes 60..63 idle); a wave walks 6 CONSECUTIVE (b, t, node) windows, whose
// outputs are one contiguous 2400-byte stretch of feat_std: the log amplitudes leave through a wave-private LDS tile as full
// 16-byte pieces.  Wave-private LDS (11.3 KB), wave-level syncs only, no workgroup barrier.
struct cplx { double re, im; };
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
// forward 5-point DFT (kernel e^{-2 pi i n k / 5})
__device__ __forceinline__ void dft5(const cplx (&x)[5], cplx (&X)[5]) {
    constexpr double C1 = 0.30901699437494742410, C2 = -0.80901699437494742410;    // cos(2 pi / 5), cos(4 pi / 5)
    constexpr double S1 = 0.95105651629515357212, S2 = 0.58778525229247312917;     // sin(2 pi / 5), sin(4 pi / 5)
    const cplx s1 = cadd(x[1], x[4]), s2 = cadd(x[2], x[3]), d1 = csub(x[1], x[4]), d2 = csub(x[2], x[3]);
    X[0] = {x[0].re + s1.re + s2.re, x[0].im + s1.im + s2.im};
    const cplx a1 = {x[0].re + C1 * s1.re + C2 * s2.re, x[0].im + C1 * s1.im + C2 * s2.im};
    const cplx a2 = {x[0].re + C2 * s1.re + C1 * s2.re, x[0].im + C2 * s1.im + C1 * s2.im};
    const cplx b1 = {S1 * d1.re + S2 * d2.re, S1 * d1.im + S2 * d2.im};
    const cplx b2 = {S2 * d1.re - S1 * d2.re, S2 * d1.im - S1 * d2.im};
    X[1] = {a1.re + b1.im, a1.im - b1.re};        // a1 - i b1
    X[4] = {a1.re - b1.im, a1.im + b1.re};        // a1 + i b1
    X[2] = {a2.re + b2.im, a2.im - b2.re};
    X[3] = {a2.re - b2.im, a2.im + b2.re};
}
// forward 10-point DFT as a 2 x 5 prime-factor transform: n = 5 n1 + 2 n2, k = 5 k1 + 6 k2 (mod 10) => W10^{nk} = (-1)^{n1 k1} W5^{n2 k2}
__device__ __forceinline__ void dft10(const cplx (&x)[10], cplx (&X)[10]) {
    cplx u[5], y0[5], y1[5];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) u[n2] = x[(2 * n2) % 10];
    dft5(u, y0);
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) u[n2] = x[(5 + 2 * n2) % 10];
    dft5(u, y1);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) {
        X[(6 * k2) % 10] = cadd(y0[k2], y1[k2]);
        X[(5 + 6 * k2) % 10] = csub(y0[k2], y1[k2]);
    }
}

#ifndef EEG_FFT_MINW
#define EEG_FFT_MINW 2          // waves per SIMD the kernel is compiled for (176 registers; 3 = 168 + 9 spilled: measured)
#endif
constexpr int kFftWin = 200, kFftBins = 100, kFftPerWave = 6, kFftWgPerCu = EEG_FFT_MINW;
constexpr int kFftRow = 11;                                              // transposed tile: 16-byte entries, row stride 11 (conflict-free)
constexpr int kFftWaveDoubles = kFftPerWave * 10 * kFftRow * 2;          // 1320 doubles = 10.3 KB per wave (the Z and output tiles alias it)

__global__ __launch_bounds__(256, EEG_FFT_MINW) void fft200_features_kernel(const float* __restrict__ raw, int N, int T, long long n_windows,
                                                              const int* __restrict__ perm, const float* __restrict__ log_scale,
                                                              float mean, float inv_std, float* __restrict__ feat_raw,
                                                              float* __restrict__ feat_std) {
    EEG_DYN_SMEM(sm);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    cplx* tile = reinterpret_cast<cplx*>(sm) + (size_t)wave * (kFftWaveDoubles / 2);
    float* ftile = reinterpret_cast<float*>(tile);                       // [6][100] log amplitudes (aliases the tile, behind a wave sync)
    const int w = lane / 10, q = lane - 10 * w;                          // window of the wave, position (n2 in stage 1, k1 in stage 2)
    const bool active = w < kFftPerWave;
    // twiddles of this lane: stage 1 -> 2: W100^{q k1}, k1 = 0..9; real-input split: W200^{q + 10 k2}, k2 = 0..9
    cplx tw1[10], tw2[10];
    {
        // two sincos per lane; the other 18 twiddles by complex rotation (error ~10 eps; twenty library sincos calls were 3 000
        // instructions = 5 us in front of every wave's first window)
        double sn, cs;
        sincos(6.283185307179586476925286766559 * (double)q / 100.0, &sn, &cs);
        const cplx r1 = {cs, -sn};                                       // W100^q
        sincos(6.283185307179586476925286766559 * (double)q / 200.0, &sn, &cs);
        const cplx r2 = {cs, sn};                                        // (cos, sin) of 2 pi q / 200
        const cplx s2 = {0.95105651629515357212, 0.30901699437494742410};   // (cos, sin) of 2 pi 10 / 200
        tw1[0] = {1.0, 0.0};
        tw2[0] = r2;
#pragma unroll
        for (int j = 1; j < 10; ++j) {
            tw1[j] = cmul(tw1[j - 1], r1);
            tw2[j] = cmul(tw2[j - 1], s2);                               // (cos, sin) of 2 pi (q + 10 j) / 200: the split uses both signs explicitly
        }
    }
    const double log_floor = log(1e-8);                                  // computeFFT: amp == 0 -> 1e-8
    const long long n_items = (n_windows + kFftPerWave - 1) / kFftPerWave;
    for (long long item = (long long)blockIdx.x * 4 + wave; item < n_items; item += (long long)gridDim.x * 4) {
        const long long w0 = item * kFftPerWave, wg = w0 + w;
        const bool live = active && wg < n_windows;
        cplx a[10], A[10];
        if (live) {
            const int nd = (int)(wg % N);
            const long long bt = wg / N;
            const int t = (int)(bt % T), b = (int)(bt / T);
            const int src = perm_source(perm, b, N, nd);     // EEG_seq_reflect[:, pair] = EEG_seq[:, swapped pair]
            const float* sig = raw + (((size_t)b * N + src) * T + t) * kFftWin;
#pragma unroll
            for (int n1 = 0; n1 < 10; ++n1) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(sig + 20 * n1 + 2 * q);        // z[10 n1 + q]
                a[n1] = {(double)v[0], (double)v[1]};
            }
            dft10(a, A);                                                 // over n1 -> k1
#pragma unroll
            for (int k1 = 0; k1 < 10; ++k1) tile[(w * 10 + q) * kFftRow + k1] = cmul(A[k1], tw1[k1]);
        }
        EEG_WAVE_SYNC();
        if (live) {
#pragma unroll
            for (int n2 = 0; n2 < 10; ++n2) a[n2] = tile[(w * 10 + n2) * kFftRow + q];
            dft10(a, A);                                                 // over n2 -> k2: A[k2] = Z[q + 10 k2]
        }
        EEG_WAVE_SYNC();
        if (live) {
#pragma unroll
            for (int k2 = 0; k2 < 10; ++k2) tile[w * 110 + q + 10 * k2] = A[k2];              // Z[k] at [w][k]
        }
        EEG_WAVE_SYNC();
        float v[10];
        if (live) {
#pragma unroll
            for (int k2 = 0; k2 < 10; ++k2) {
                const int k = q + 10 * k2;
                const cplx zp = tile[w * 110 + (k == 0 ? 0 : 100 - k)];
                const double ar = A[k2].re + zp.re, ai = A[k2].im - zp.im;                    // Z[k] + conj Z[100-k]
                const double br = A[k2].re - zp.re, bi = A[k2].im + zp.im;                    // Z[k] - conj Z[100-k]
                const double xr = ar + (tw2[k2].re * bi - tw2[k2].im * br);
                const double xi = ai - (tw2[k2].re * br + tw2[k2].im * bi);
                const double pw = 0.25 * (xr * xr + xi * xi);
                // log|X| = ln2/2 * (e + log2 m), |X|^2 = m 2^e with m in [0.5, 1): the exponent split is exact in fp64, the mantissa's
                // log2 is one v_log_f32 (absolute error ~1e-7 on a value in (-1, 0]) and the two parts are combined in fp64 -- the fp64
                // library log (~70 fp64 instructions per bin) was 60 % of this kernel's instructions
                int ex;
                const double mant = frexp(pw, &ex);
                const double lg = 0.34657359027997264 * ((double)ex + (double)fast_log2((float)mant));
                v[k2] = (float)(pw == 0.0 ? log_floor : lg);
            }
        }
        EEG_WAVE_SYNC();                                                 // every partner read is done: the tile becomes the output tile
        if (live) {
#pragma unroll
            for (int k2 = 0; k2 < 10; ++k2) ftile[w * kFftBins + q + 10 * k2] = v[k2];
        }
        EEG_WAVE_SYNC();
        // 6 windows x 100 floats leave as 16-byte pieces: feat_std at the window's own (b, t, node) slot (6 windows = one 2400-byte
        // stretch), feat_raw at the SOURCE node's slot (perm is a permutation: every slot is written once)
        for (int c = lane; c < kFftPerWave * kFftBins / 4; c += 64) {
            const int j = c / 25, pos = 4 * (c - 25 * j);
            const long long wj = w0 + j;
            if (wj >= n_windows) continue;
            const f32x4 val = *reinterpret_cast<const f32x4*>(ftile + 4 * c);
            const int nd = (int)(wj % N);
            const long long bt = wj / N;
            const int b = (int)(bt / T);
            if (feat_raw != nullptr) {
                const int src = perm_source(perm, b, N, nd);
                *reinterpret_cast<f32x4*>(feat_raw + ((size_t)bt * N + src) * kFftBins + pos) = val;
            }
            if (feat_std != nullptr) {
                const double ls = log_scale != nullptr ? (double)log_scale[b] : 0.0;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (float)((((double)val[e] + ls) - (double)mean) * (double)inv_std);
                *reinterpret_cast<f32x4*>(feat_std + (size_t)wj * kFftBins + pos) = o;
            }
        }
        EEG_WAVE_SYNC();                                                 // the output tile is free again
    }
}

// Data augmentation drawn on the device (dataloader_detection.py:384-389: `_random_reflect` then `_random_scale` per sample, in the
// DataLoader workers): clip b takes Philox counter used[1] + b of the {seed, offset} pair `used` (eeg_dcrnn_rng_take): word 0 is the
// fair coin of `np.random.choice([True, False])` (top bit), word 1 the uniform of `np.random.uniform(0.8, 1.2)`.  Outputs, all
// read by later launches of the same step: flags[b]; perm[b][n] = the source channel of node n (swap_perm[n] if reflected, else n)
// and log_scale[b] = log(scale) -- the `perm` / `log_scale` operands of the featurisation kernels above; and, for the distance
// graph (`_get_combined_graph(swap_nodes)`, :405-409), the per-clip supports S_out[s][b] = flags[b] ? S_refl[s] : S_plain[s].
__global__ __launch_bounds__(128) void augment_draw_kernel(const unsigned long long* __restrict__ used, int N, const int* __restrict__ swap_perm,
                                                            int* __restrict__ flags, int* __restrict__ perm, float* __restrict__ log_scale,
                                                            const float* __restrict__ S_plain, const float* __restrict__ S_refl, int nsup,
                                                            float* __restrict__ S_out) {
    const int b = blockIdx.x, B = gridDim.x;
    const unsigned long long seed = used[0], c = used[1] + (unsigned long long)b;
    unsigned w[4];
    philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
    const int reflect = (int)(w[0] >> 31);
    if (threadIdx.x == 0) {
        flags[b] = reflect;
        const double scale = 0.8 + 0.4 * ((double)w[1] * (1.0 / 4294967296.0));
        log_scale[b] = (float)log(scale);
    }
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int s = swap_perm[n];
        perm[b * N + n] = (reflect && (unsigned)s < (unsigned)N) ? s : n;
    }
    if (S_out != nullptr) {
        const int NN = N * N;
        for (int e = threadIdx.x; e < nsup * NN; e += blockDim.x) {
            const int s = e / NN, r = e - s * NN;
            S_out[((size_t)s * B + b) * NN + r] = reflect ? S_refl[e] : S_plain[e];
        }
    }
}

}  // namespace eeg
