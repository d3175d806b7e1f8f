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
        // a label outside 0..C-1 (torch: a device-side assert that ends the process) makes the LOSS NaN instead of reading logits out
        // of bounds; the gradient of that clip is the softmax alone
        const long long tl = y[i];
        const bool t_ok = tl >= 0 && tl < (long long)C;
        const int t = t_ok ? (int)tl : -1;
        acc += lse - (t_ok ? r[t] : __builtin_nanf(""));
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

// utils.compute_regression_loss (utils.py:431-495): optional scalar StandardScaler inverse transform
// (v*std + mean), mask = (y_true != mask_val), loss = sum(e * mask) / count(mask) with e = |d| (kind 0,
// masked MAE) or d^2 followed by sqrt (kind 1: `masked_mse_loss`, really a masked RMSE).
// stage 1: per-block partial sums [sum | count]; stage 2 (one block): loss value + gradient scale;
// stage 3: dpred.  Fixed-order reductions.
constexpr int kLossBlocks = 1024;
__device__ __forceinline__ void masked_terms(float p, float y, float mean, float std_, int scaled, float mask_val,
                                             float& d, float& mk) {
    // two roundings like the reference's `data * std + mean` (no FMA contraction: the mask tests ys against mask_val)
    const float ps = scaled ? unfused_mul_add(p, std_, mean) : p, ys = scaled ? unfused_mul_add(y, std_, mean) : y;
    d = ps - ys;
    mk = ys != mask_val ? 1.f : 0.f;
}
// 16-byte loads, two per operand in flight per thread; the n % 4 tail elements go to the last thread of the grid.
__global__ __launch_bounds__(256) void masked_loss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                                  size_t n, float mean, float std_, int scaled,
                                                                  float mask_val, int kind, float* __restrict__ part) {
    EEG_DYN_SMEM(sm);                                   // [2][256]
    float s = 0.f, c = 0.f;
    auto term = [&](float p, float yv) {
        float d, mk;
        masked_terms(p, yv, mean, std_, scaled, mask_val, d, mk);
        s += (kind == 0 ? fabsf(d) : d * d) * mk;
        c += mk;
    };
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const f32x4 p0 = ld4(pred + 4 * i), y0 = ld4(y + 4 * i), p1 = ld4(pred + 4 * (i + stride)), y1 = ld4(y + 4 * (i + stride));
#pragma unroll
        for (int r = 0; r < 4; ++r) term(p0[r], y0[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) term(p1[r], y1[r]);
    }
    if (i < n4) {
        const f32x4 p0 = ld4(pred + 4 * i), y0 = ld4(y + 4 * i);
#pragma unroll
        for (int r = 0; r < 4; ++r) term(p0[r], y0[r]);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == blockDim.x - 1)
        for (size_t e = 4 * n4; e < n; ++e) term(pred[e], y[e]);
    sm[threadIdx.x] = s;
    sm[256 + threadIdx.x] = c;
    __syncthreads();
    for (int k = blockDim.x / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { sm[threadIdx.x] += sm[threadIdx.x + k]; sm[256 + threadIdx.x] += sm[256 + threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x] = sm[0]; part[kLossBlocks + blockIdx.x] = sm[256]; }
}
// part[2*kLossBlocks] = loss, [+1] = gradient scale (MAE: 1/count; RMSE: 1/(count*loss)); 0 when count == 0
// one block of 256 threads: thread i adds parts i, i + 256, ...; then a fixed tree.
__global__ __launch_bounds__(256) void masked_loss_finish_kernel(float* __restrict__ part, int nblk, int kind, float* __restrict__ loss) {
    EEG_DYN_SMEM(sm);                                   // [2][256]
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) { s += part[i]; c += part[kLossBlocks + i]; }
    sm[threadIdx.x] = s;
    sm[256 + threadIdx.x] = c;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { sm[threadIdx.x] += sm[threadIdx.x + k]; sm[256 + threadIdx.x] += sm[256 + threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    s = sm[0];
    c = sm[256];
    float l = 0.f, sc = 0.f;
    if (c > 0.f) {
        l = kind == 0 ? s / c : sqrtf(s / c);
        sc = kind == 0 ? 1.f / c : (l > 0.f ? 1.f / (c * l) : 0.f);
    }
    loss[0] = l;
    part[2 * kLossBlocks] = l;
    part[2 * kLossBlocks + 1] = sc;
}
__global__ __launch_bounds__(256) void masked_loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                               size_t n, float mean, float std_, int scaled, float mask_val,
                                                               int kind, const float* __restrict__ part, float* __restrict__ dpred) {
    const float sc = part[2 * kLossBlocks + 1] * (scaled ? std_ : 1.f);
    auto grad = [&](float p, float yv) {
        float d, mk;
        masked_terms(p, yv, mean, std_, scaled, mask_val, d, mk);
        const float e = kind == 0 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : d;
        return e * mk * sc;
    };
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 p0 = ld4(pred + 4 * i), y0 = ld4(y + 4 * i);
        f32x4 g;
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = grad(p0[r], y0[r]);
        st4(dpred + 4 * i, g);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == blockDim.x - 1)
        for (size_t e = 4 * n4; e < n; ++e) dpred[e] = grad(pred[e], y[e]);
}

// p[0 .. n16) <- 0 in 16-byte pieces, then the < 16 trailing bytes (optimizer.zero_grad() on the flat bucket; a memset node of the
// runtime costs ~4.7 us per call in a replayed graph, this kernel ~2 us)
__global__ __launch_bounds__(256) void zero_kernel(float4* __restrict__ p, size_t n16, unsigned char* __restrict__ tail, int ntail) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

// stage 1: per-block partial sums of g^2 (fixed assignment of elements to blocks/threads)
// step_dev (nullable): the optimiser's step counter on the device, advanced here -- one launch AHEAD of clip_adam_kernel, every
// block of which reads it -- so that a captured graph of the whole step counts its own replays
__global__ void sqnorm_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part, int* __restrict__ step_dev) {
    EEG_DYN_SMEM(sm);
    if (step_dev != nullptr && blockIdx.x == 0 && threadIdx.x == 0) step_dev[0] += 1;
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
                                 float bc1, float bc2_sqrt, float grad_scale, float* __restrict__ norm_out,
                                 const int* __restrict__ step_dev, const float* __restrict__ lr_dev) {
    if (step_dev != nullptr) {      // step count and learning rate live on the device (eeg_dcrnn_clip_adam_dev): the bias corrections
        const int step = step_dev[0];   // are formed here, in fp64 like torch.optim.Adam's Python scalars
        bc1 = (float)(1.0 - pow((double)b1, (double)step));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
        lr = lr_dev[0];
    }
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
