// Training-step tail of train.py:203-206,266-275 on the flat parameter / gradient buffers:
// losses that seed backward (BCE-with-logits, cross-entropy; forward value + dlogits in one launch),
// squared gradient norm (fixed-order two-stage reduction) and clip_grad_norm_ + Adam (coupled L2)
// fused into one pass.  Replaces ~20 tiny framework kernels and their host round trips.
#pragma once
#include "common.h"

namespace eeg {

// nn.BCEWithLogitsLoss() (mean) on logits (B,), targets y (B,): loss[0], dlogits = (sigmoid(x)-y)/B.
// single workgroup (B is a batch size), fixed-order tree reduction.
__global__ void bce_logits_kernel(const float* __restrict__ x, const float* __restrict__ y, int B,
                                  float* __restrict__ loss, float* __restrict__ dx) {
    EEG_DYN_SMEM(sm);
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float v = x[i], t = y[i];
        acc += fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
        dx[i] = (1.f / (1.f + expf(-v)) - t) / (float)B;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = sm[0] / (float)B;
}

// nn.CrossEntropyLoss() (mean) on logits (B,C), integer targets (B,): loss[0], dlogits = (softmax - onehot)/B
__global__ void ce_logits_kernel(const float* __restrict__ x, const long long* __restrict__ y, int B, int C,
                                 float* __restrict__ loss, float* __restrict__ dx) {
    EEG_DYN_SMEM(sm);
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float* r = x + (size_t)i * C;
        float mx = r[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, r[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(r[c] - mx);
        const float lse = mx + logf(se);
        const int t = (int)y[i];
        acc += lse - r[t];
        for (int c = 0; c < C; ++c) dx[(size_t)i * C + c] = (expf(r[c] - lse) - (c == t ? 1.f : 0.f)) / (float)B;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = sm[0] / (float)B;
}

// stage 1: per-block partial sums of g^2 (fixed assignment of elements to blocks/threads)
__global__ void sqnorm_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part) {
    EEG_DYN_SMEM(sm);
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc = fmaf(g[i], g[i], acc);
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

// stage 2 + clip + Adam: every block re-reduces the (few) partials in the same fixed order, then
// updates its slice:  g <- g * min(1, max_norm/(||g||+1e-6));  torch.optim.Adam (coupled weight
// decay):  g += wd*p; m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g^2; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2)+eps)
__global__ void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, size_t n, const float* __restrict__ part, int nparts,
                                 float max_norm, float lr, float b1, float b2, float eps, float wd,
                                 float bc1, float bc2_sqrt, float grad_scale, float* __restrict__ norm_out) {
    float tot = 0.f;
    for (int i = 0; i < nparts; ++i) tot += part[i];
    const float norm = sqrtf(tot) * grad_scale;        // grad_scale: 1/world after a summed all-reduce
    float clip = max_norm / (norm + 1e-6f);
    clip = clip < 1.f ? clip : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out != nullptr) norm_out[0] = norm;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        float gi = g[i] * grad_scale * clip;
        g[i] = gi;
        gi = fmaf(wd, pi, gi);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

}  // namespace eeg
