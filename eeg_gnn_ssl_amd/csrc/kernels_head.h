// Classification / detection head of model/model.py:260-270 (gather at len-1, relu, per-node
// Linear(H->C), max over nodes) and its backward.  Negligible work; one small launch each.
#pragma once
#include "common.h"

namespace eeg {

// last[b][:] = Htop[lengths[b]-1][b][:]            (utils.py:346-357, batch-first gather)
__global__ void gather_last_kernel(const float* __restrict__ Htop, const long long* __restrict__ lengths,
                                   int T, int B, int NH, float* __restrict__ last) {
    const size_t total = (size_t)B * NH;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = i / NH, e = i % NH;
        int t = (int)lengths[b] - 1;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        last[i] = Htop[((size_t)t * B + b) * NH + e];
    }
}

// logits[b][c] = max_n ( sum_h relu(z[b][n][h]) W[c][h] + bias[c] ); arg = first maximising node.
// one 64-thread workgroup per clip; thread n (< N) scores node n for every class.
__global__ void cls_head_fwd_kernel(const float* __restrict__ z, const float* __restrict__ W,
                                    const float* __restrict__ bias, int B, int N, int H, int C,
                                    float* __restrict__ logits, int* __restrict__ arg) {
    EEG_DYN_SMEM(sm);                   // [N][C] node logits
    const int b = blockIdx.x, n = threadIdx.x;
    if (n < N) {
        for (int c = 0; c < C; ++c) {
            float s = bias[c];
            for (int h = 0; h < H; ++h) s = fmaf(fmaxf(z[((size_t)b * N + n) * H + h], 0.f), W[c * H + h], s);
            sm[n * C + c] = s;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float best = sm[c];
        int bi = 0;
        for (int q = 1; q < N; ++q)
            if (sm[q * C + c] > best) { best = sm[q * C + c]; bi = q; }
        logits[(size_t)b * C + c] = best;
        arg[(size_t)b * C + c] = bi;
    }
}

// dz[b][n][h] = sum_c [arg[b][c]==n] dlogits[b][c] W[c][h] * (z>0)
__global__ void cls_head_bwd_dz_kernel(const float* __restrict__ z, const float* __restrict__ W,
                                       const float* __restrict__ dlogits, const int* __restrict__ arg,
                                       int B, int N, int H, int C, float* __restrict__ dz) {
    const size_t total = (size_t)B * N * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int h = i % H, n = (i / H) % N, b = i / ((size_t)H * N);
        float s = 0.f;
        if (z[i] > 0.f)
            for (int c = 0; c < C; ++c)
                if (arg[(size_t)b * C + c] == n) s = fmaf(dlogits[(size_t)b * C + c], W[c * H + h], s);
        dz[i] = s;
    }
}

// dW[c][h] = sum_b dlogits[b][c] relu(z[b][arg[b][c]][h]);  dbias[c] = sum_b dlogits[b][c]
// block = 16 outputs x 16 batch slices, LDS combine in a fixed order (deterministic).
__global__ void cls_head_bwd_w_kernel(const float* __restrict__ z, const float* __restrict__ dlogits,
                                      const int* __restrict__ arg, int B, int N, int H, int C,
                                      float* __restrict__ dW, float* __restrict__ dbias) {
    EEG_DYN_SMEM(sm);                                 // [16][16]
    const int o = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    float s = 0.f;
    if (i < C * H) {
        const int c = i / H, h = i % H;
        for (int b = q; b < B; b += 16) {
            const int n = arg[(size_t)b * C + c];
            s = fmaf(dlogits[(size_t)b * C + c], fmaxf(z[((size_t)b * N + n) * H + h], 0.f), s);
        }
    } else if (i < C * H + C) {
        const int c = i - C * H;
        for (int b = q; b < B; b += 16) s += dlogits[(size_t)b * C + c];
    }
    sm[q * 16 + o] = s;
    __syncthreads();
    if (q == 0 && i < C * H + C) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += sm[k * 16 + o];
        if (i < C * H) dW[i] = t; else dbias[i - C * H] = t;
    }
}

}  // namespace eeg
