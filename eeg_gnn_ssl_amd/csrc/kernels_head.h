// Classification / detection head of model/model.py:260-270 (gather at len-1, relu, per-node
// Linear(H->C), max over nodes) and its backward.  Negligible work; one small launch each.
#pragma once
#include "common.h"

namespace eeg {

__device__ __forceinline__ f32x4 ld4g(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return (f32x4){v.x, v.y, v.z, v.w};
}

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

// logits[b][c] = max_n ( sum_h relu(drop(z)[b][n][h]) W[c][h] + bias[c] ); arg = first maximising node.
// drop = nn.Dropout(p) of model.py:267 (training): element kept with probability 1-p and scaled by 1/(1-p); since the scale is
// positive, relu(drop(z)) = mask * relu(z).  One 64-thread workgroup per clip: the masked relu(z) rows are staged in LDS (one
// Philox call per 16-byte group), then thread n (< N) scores node n for every class.
// used (drop.on only): device {seed, offset} of this call (common.h rng_take_kernel); element e of z takes word e % 4 of counter
// offset + e / 4.
__global__ void cls_head_fwd_kernel(const float* __restrict__ z, const float* __restrict__ W,
                                    const float* __restrict__ bias, int B, int N, int H, int C, DropCfg drop,
                                    const unsigned long long* __restrict__ used, float* __restrict__ logits, int* __restrict__ arg) {
    EEG_DYN_SMEM(sm);                   // [N][C] node logits, then [N][H] masked relu(z)
    float* zs = sm + N * C;
    const int b = blockIdx.x, n = threadIdx.x;
    const unsigned long long seed = drop.on ? used[0] : 0ull, off = drop.on ? used[1] : 0ull;
    const int groups = N * H / 4;       // H % 4 == 0
    for (int g = threadIdx.x; g < groups; g += blockDim.x) {
        const size_t gg = (size_t)b * groups + g;
        f32x4 v = ld4g(z + 4 * gg);
        f32x4 m = {1.f, 1.f, 1.f, 1.f};
        if (drop.on) m = dropout_mask4(seed, off, gg, drop.thr, drop.scale);
#pragma unroll
        for (int j = 0; j < 4; ++j) zs[4 * g + j] = fmaxf(v[j], 0.f) * m[j];
    }
    __syncthreads();
    if (n < N) {
        for (int c = 0; c < C; ++c) {
            float s = bias[c];
            for (int h = 0; h < H; ++h) s = fmaf(zs[n * H + h], W[c * H + h], s);
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

// the mask x scale value of element i of the (B,N,H) head input (1 without dropout)
__device__ __forceinline__ float head_mask(const DropCfg& drop, const unsigned long long* __restrict__ used, size_t i) {
    if (!drop.on) return 1.f;
    return dropout_mask4(used[0], used[1], i >> 2, drop.thr, drop.scale)[i & 3];
}

// dz[b][n][h] = sum_c [arg[b][c]==n] dlogits[b][c] W[c][h] * (z>0) * mask
__global__ void cls_head_bwd_dz_kernel(const float* __restrict__ z, const float* __restrict__ W,
                                       const float* __restrict__ dlogits, const int* __restrict__ arg,
                                       int B, int N, int H, int C, DropCfg drop, const unsigned long long* __restrict__ used,
                                       float* __restrict__ dz) {
    const size_t total = (size_t)B * N * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int h = i % H, n = (i / H) % N, b = i / ((size_t)H * N);
        float s = 0.f;
        if (z[i] > 0.f)
            for (int c = 0; c < C; ++c)
                if (arg[(size_t)b * C + c] == n) s = fmaf(dlogits[(size_t)b * C + c], W[c * H + h], s);
        dz[i] = s * head_mask(drop, used, i);
    }
}

// dW[c][h] = sum_b dlogits[b][c] mask relu(z[b][arg[b][c]][h]);  dbias[c] = sum_b dlogits[b][c]
// block = 16 outputs x 16 batch slices, LDS combine in a fixed order (deterministic).
__global__ void cls_head_bwd_w_kernel(const float* __restrict__ z, const float* __restrict__ dlogits,
                                      const int* __restrict__ arg, int B, int N, int H, int C, DropCfg drop,
                                      const unsigned long long* __restrict__ used, float* __restrict__ dW,
                                      float* __restrict__ dbias) {
    EEG_DYN_SMEM(sm);                                 // [16][16]
    const int o = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    float s = 0.f;
    if (i < C * H) {
        const int c = i / H, h = i % H;
        for (int b = q; b < B; b += 16) {
            const int n = arg[(size_t)b * C + c];
            const size_t e = ((size_t)b * N + n) * H + h;
            s = fmaf(dlogits[(size_t)b * C + c], fmaxf(z[e], 0.f) * head_mask(drop, used, e), s);
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

// ---- head + criterion + their backward in one launch (train.py:266-272 of an optimisation step) ------------------------------------
// cls_head_fwd_kernel -> bce_logits_kernel / ce_logits_kernel (kernels_tail.h) -> cls_head_bwd_dz_kernel -> cls_head_bwd_w_kernel,
// fused for the training step: one WAVE per clip (4 clips per workgroup) stages mask * relu(z) of its clip in LDS once and derives
// everything from it -- logits / arg (kept: evaluation metrics and the parity tests read them), the clip's loss term, dlogits
// (mean reduction: 1/B folded in), dz (the seed of the BPTT kernels; rows of nodes that are not an arg-max are zero) and the clip's
// contribution to dW / dbias.  The contributions and loss terms of a workgroup's clips are summed in LDS in clip order and written
// as partial[blk][O + 1] (O = C*H + C); cls_head_loss_finish_kernel adds the partials in block order (fixed order: bit-reproducible).
// kind 0: nn.BCEWithLogitsLoss on logits (B,1), float targets; kind 1: nn.CrossEntropyLoss on (B,C), int64 targets (a label outside
// 0..C-1 makes the LOSS NaN, see ce_logits_kernel).  LDS per wave: [N][C] node logits | [N][H] masked relu(z) | [C] dlogits | [C] arg.
__host__ __device__ inline int cls_tail_wave_floats(int N, int H, int C) { return N * C + N * H + 2 * C; }
__global__ __launch_bounds__(256) void cls_head_loss_kernel(const float* __restrict__ z, const float* __restrict__ W, const float* __restrict__ bias,
                                                            const void* __restrict__ targets, int kind, int B, int N, int H, int C, DropCfg drop,
                                                            const unsigned long long* __restrict__ used, float* __restrict__ logits,
                                                            int* __restrict__ arg, float* __restrict__ dlogits, float* __restrict__ dz,
                                                            float* __restrict__ partial) {
    EEG_DYN_SMEM(sm);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x * 4 + w, O = C * H + C;
    float* nl = sm + w * cls_tail_wave_floats(N, H, C);       // [N][C]
    float* zs = nl + N * C;                                   // [N][H]
    float* dl = zs + N * H;                                   // [C]
    int* ag = reinterpret_cast<int*>(dl + C);                 // [C]
    float* contrib = sm + 4 * cls_tail_wave_floats(N, H, C);  // [4][O + 1]
    const bool live = b < B;
    const int groups = N * H / 4;
    if (live) {
        const unsigned long long seed = drop.on ? used[0] : 0ull, off = drop.on ? used[1] : 0ull;
        for (int g = lane; g < groups; g += 64) {
            const size_t gg = (size_t)b * groups + g;
            const f32x4 v = ld4g(z + 4 * gg);
            f32x4 m = {1.f, 1.f, 1.f, 1.f};
            if (drop.on) m = dropout_mask4(seed, off, gg, drop.thr, drop.scale);
#pragma unroll
            for (int j = 0; j < 4; ++j) zs[4 * g + j] = fmaxf(v[j], 0.f) * m[j];
        }
        EEG_WAVE_SYNC();
        if (lane < N) {
            for (int c = 0; c < C; ++c) {
                float s = bias[c];
                for (int h = 0; h < H; ++h) s = fmaf(zs[lane * H + h], W[c * H + h], s);
                nl[lane * C + c] = s;
            }
        }
        EEG_WAVE_SYNC();
        for (int c = lane; c < C; c += 64) {
            float best = nl[c];
            int bi = 0;
            for (int q = 1; q < N; ++q)
                if (nl[q * C + c] > best) { best = nl[q * C + c]; bi = q; }
            logits[(size_t)b * C + c] = best;
            arg[(size_t)b * C + c] = bi;
            dl[c] = best;                                      // (the logit for now)
            ag[c] = bi;
        }
        EEG_WAVE_SYNC();
        if (lane == 0) {                                       // the clip's loss term and d loss / d logits (C is a handful)
            float term;
            if (kind == 0) {
                const float v = dl[0], t = reinterpret_cast<const float*>(targets)[b];
                term = fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
                dl[0] = (1.f / (1.f + expf(-v)) - t) / (float)B;
            } else {
                float mx = dl[0];
                for (int c = 1; c < C; ++c) mx = fmaxf(mx, dl[c]);
                float se = 0.f;
                for (int c = 0; c < C; ++c) se += expf(dl[c] - mx);
                const float lse = mx + logf(se);
                const long long tl = reinterpret_cast<const long long*>(targets)[b];
                const bool t_ok = tl >= 0 && tl < (long long)C;
                const int t = t_ok ? (int)tl : -1;
                term = lse - (t_ok ? dl[t] : __builtin_nanf(""));
                for (int c = 0; c < C; ++c) dl[c] = (expf(dl[c] - lse) - (c == t ? 1.f : 0.f)) / (float)B;
            }
            contrib[w * (O + 1) + O] = term;
            for (int c = 0; c < C; ++c) dlogits[(size_t)b * C + c] = dl[c];
        }
        EEG_WAVE_SYNC();
        // dz[b][n][h] = sum_c [arg_c == n] dl_c W[c][h] * (z > 0) * mask; zs > 0 <=> z > 0 and kept, and the mask value is then drop.scale
        const float keep = drop.on ? drop.scale : 1.f;
        for (int g = lane; g < groups; g += 64) {
            const int n = (4 * g) / H, h0 = (4 * g) % H;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < C; ++c)
                if (ag[c] == n) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(dl[c], W[c * H + h0 + j], o[j]);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = zs[4 * g + j] > 0.f ? o[j] * keep : 0.f;
            *reinterpret_cast<float4*>(dz + ((size_t)b * groups + g) * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
        // this clip's dW[c][h] = dl_c * mask relu(z)[arg_c][h], dbias[c] = dl_c
        for (int i = lane; i < O; i += 64) {
            float v;
            if (i < C * H) { const int c = i / H, h = i % H; v = dl[c] * zs[ag[c] * H + h]; }
            else v = dl[i - C * H];
            contrib[w * (O + 1) + i] = v;
        }
    }
    __syncthreads();
    const int nlive = B - blockIdx.x * 4 < 4 ? B - blockIdx.x * 4 : 4;
    for (int i = threadIdx.x; i <= O; i += blockDim.x) {
        float s = 0.f;
        for (int q = 0; q < nlive; ++q) s += contrib[q * (O + 1) + i];
        partial[(size_t)blockIdx.x * (O + 1) + i] = s;
    }
}
// dW / dbias / loss from the per-workgroup partials, in block order
__global__ __launch_bounds__(256) void cls_head_loss_finish_kernel(const float* __restrict__ partial, int nblk, int B, int H, int C,
                                                                   float* __restrict__ dW, float* __restrict__ dbias, float* __restrict__ loss) {
    const int O = C * H + C;
    for (int i = threadIdx.x; i <= O; i += blockDim.x) {
        float s = 0.f;
        int q = 0;
        for (; q + 8 <= nblk; q += 8) {                        // eight loads in flight, added in order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(q + u) * (O + 1) + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; q < nblk; ++q) s += partial[(size_t)q * (O + 1) + i];
        if (i < C * H) dW[i] = s;
        else if (i < O) dbias[i - C * H] = s;
        else loss[0] = s / (float)B;
    }
}

// ---- dropout generator plumbing (common.h: Philox4x32-10 keep masks) --------------------------------------------------------
// first launch of a forward entry point that drops: hands the {seed, offset} pair of this call to `used` and advances the
// generator state by the counters the call will draw.  The compute kernels behind it on the stream only read `used`.
__global__ void rng_take_kernel(unsigned long long* __restrict__ state, unsigned long long* __restrict__ used, unsigned long long groups) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long seed = state[0], off = state[1];
        used[0] = seed;
        used[1] = off;
        state[1] = off + groups;
    }
}
// Scheduled sampling on the device (model.py:194-200 + utils.py:385-390 `compute_sampling_threshold`): the reference draws one
// `random.random() < k / (k + exp(batches_seen / k))` per decoder step on the host; here the T flags of a forward call come from the
// same Philox generator as the dropout masks (flag t = word t % 4 of counter offset + t / 4, u = word / 2^32 in fp64), and both the
// generator offset and the `samples_seen` counter advance ON THE STREAM -- a captured training step replays with a fresh draw and
// the decayed threshold.  seen[0] (int64) is read, then incremented by `inc` (the global batch: train_ssl.py:178 `step += batch_size`).
__global__ void teacher_flags_kernel(unsigned long long* __restrict__ state, long long* __restrict__ seen, long long inc,
                                     double decay_steps, int T, int* __restrict__ flags) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const unsigned long long seed = state[0], off = state[1];
    const long long n = seen[0];
    const double ratio = decay_steps / (decay_steps + exp((double)n / decay_steps));
    for (int t0 = 0; t0 < T; t0 += 4) {
        unsigned w[4];
        const unsigned long long c = off + (unsigned long long)(t0 >> 2);
        philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
        for (int j = 0; j < 4 && t0 + j < T; ++j) flags[t0 + j] = ((double)w[j] * (1.0 / 4294967296.0) < ratio) ? 1 : 0;
    }
    state[1] = off + (unsigned long long)((T + 3) / 4);
    seen[0] = n + inc;
}
// mask[e] = keep(e) * scale for e < n: the values the fused kernels multiply with, materialised (tests hand them to the oracle)
__global__ void dropout_mask_kernel(const unsigned long long* __restrict__ used, size_t n, DropCfg drop, float* __restrict__ mask) {
    const size_t groups = (n + 3) / 4;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
        const f32x4 m = drop.on ? dropout_mask4(used[0], used[1], g, drop.thr, drop.scale) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * g + j < n) mask[4 * g + j] = m[j];
    }
}
// y = x * mask (per-step decoder path: the projection input) / x *= mask (its gradient: launched IN PLACE, x == y, hence no
// __restrict__ on the two; every thread reads its own 16 bytes before it writes them); e0 = index of x[0] in the dropped tensor
__global__ void dropout_apply_kernel(const float* x, float* y, size_t n, size_t e0,
                                     const unsigned long long* __restrict__ used, DropCfg drop) {
    const size_t groups = n / 4;        // n % 4 == 0, e0 % 4 == 0
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
        const f32x4 m = dropout_mask4(used[0], used[1], e0 / 4 + g, drop.thr, drop.scale);
        const float4 v = *reinterpret_cast<const float4*>(x + 4 * g);
        *reinterpret_cast<float4*>(y + 4 * g) = make_float4(v.x * m[0], v.y * m[1], v.z * m[2], v.w * m[3]);
    }
}

}  // namespace eeg
