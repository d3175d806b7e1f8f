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

__global__ __launch_bounds__(64) void fft_features_kernel(const float* __restrict__ raw, int N, int T, int W, int tchunk,
                                                          const int* __restrict__ perm, const float* __restrict__ log_scale,
                                                          float mean, float inv_std, float* __restrict__ feat_raw,
                                                          float* __restrict__ feat_std) {
    EEG_DYN_SMEM(sm);
    double* xs = reinterpret_cast<double*>(sm);          // [W]
    const int lane = threadIdx.x, b = blockIdx.x / N, nd = blockIdx.x % N, H2 = W / 2;
    const int src = perm != nullptr ? perm[b * N + nd] : nd;            // EEG_seq_reflect[:, pair] = EEG_seq[:, swapped pair]
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

}  // namespace eeg
