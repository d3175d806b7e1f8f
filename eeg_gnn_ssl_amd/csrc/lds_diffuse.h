// LDS-resident graph diffusion building blocks shared by the standalone diffusion kernels and
// the recurrent kernels: the (M-1) non-identity hop-polynomial matrices of one graph live in LDS
// zero-padded to 32x32 and are applied to an LDS-resident (32 x W) feature tile with fp32 MFMA
// (D[n][f] = sum_n' P_m[n][n'] X[n'][f], or P_m^T for the adjoint).
#pragma once
#include "common.h"

namespace eeg {

constexpr int kPStride = 34;                       // lds_stride(32)
constexpr int kPFloats = kMaxNodes * kPStride;     // one padded 32x32 matrix

// P (global): (Bp, M-1, N, N); graph g.  Pl (LDS): (M-1) x [32][kPStride], zero padded.
__device__ __forceinline__ void lds_load_polys(float* Pl, const float* __restrict__ P, int g, int M, int N) {
    const int total = (M - 1) * kPFloats;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int m1 = e / kPFloats, r = (e % kPFloats) / kPStride, c = e % kPStride;
        float v = 0.f;
        if (r < N && c < N) v = P[(((size_t)g * (M - 1) + m1) * N + r) * N + c];
        Pl[e] = v;
    }
}

// Apply hop matrices m = 1..M-1 to the (32 x W) source block buf[:, src_off : src_off+W) and
// write results to buf[:, dst_off + (m-1)*dst_step : +W).  Rows >= N of the source must be zero
// (or finite); rows >= N of the result are written as exact zeros (P is zero padded).
// Work items = (m, row tile, col tile), dealt round-robin to the block's waves; each wave works on
// UNR tiles at once (independent accumulators: the 16x16x4 MFMA has a 40-cycle dependent latency).
// W % 16 == 0.  Result rows >= rows_limit are not stored.
template <bool ADJ, int UNR = 4>
__device__ __forceinline__ void lds_diffuse_tiles(float* buf, int stride, int src_off, int dst_off,
                                                  int dst_step, int W, const float* Pl, int M, int N,
                                                  int rows_limit) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int nct = W / 16, nks = ceil_div(N, 4);
    const int ntiles = (M - 1) * 2 * nct;
    const int lr = lane & 15, lg = lane >> 4;
    for (int t0 = wave; t0 < ntiles; t0 += nwaves * UNR) {
        f32x4 acc[UNR];
        int aoff[UNR], boff[UNR], doff[UNR], row0[UNR];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int t = t0 + u * nwaves;
            ok[u] = t < ntiles;
            const int tt = ok[u] ? t : t0;
            const int ct = tt % nct, rt = (tt / nct) & 1, m1 = tt / (2 * nct);
            aoff[u] = m1 * kPFloats + (ADJ ? (lg * kPStride + rt * 16 + lr) : ((rt * 16 + lr) * kPStride + lg));
            boff[u] = lg * stride + src_off + ct * 16 + lr;
            row0[u] = rt * 16 + 4 * lg;
            doff[u] = row0[u] * stride + dst_off + m1 * dst_step + ct * 16 + lr;
            acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int ks = 0; ks < nks; ++ks) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const float a = Pl[aoff[u] + (ADJ ? 4 * ks * kPStride : 4 * ks)];
                const float b = buf[boff[u] + 4 * ks * stride];
                acc[u] = mfma16(a, b, acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (ok[u]) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row0[u] + r < rows_limit) buf[doff[u] + r * stride] = acc[u][r];
            }
        }
    }
}

// ---- register-resident variant used by the persistent recurrent kernels -----------------------
// The hop polynomials never change during a sequence, so each lane keeps its MFMA B-fragments in registers.
// NKS = ceil(N/4) k-steps of the node mix (5 for the 19-electrode graph, 8 for up to 32 nodes).
//   NKS == 8: one chain per (hop m, node tile rt): pf[(m-1)*2 + rt][ks]                     -> (M-1)*2 chains
//   NKS == 5 (at most 20 nodes): the second node tile holds only output nodes 16..19, so the second tiles of up
//     to FOUR hops share one chain: its 16 output slots are (hop, r) pairs, slot lr = 4*(hop % 4) + r <-> row
//     16 + r of hop plane `hop`.  Chains: M-1 first tiles + ceil((M-1)/4) packed ones -- 3 instead of 4 at M = 3,
//     5 instead of 8 at M = 5 -- and a lane of a packed chain still ends up with 4 consecutive columns of ONE
//     (hop, node) row, so its result goes out as one 16-byte write like every other.
template <int M, int NKS>
constexpr int poly_chains() { return NKS == 5 ? (M - 1) + (M - 1 + 3) / 4 : (M - 1) * 2; }
// array extent for the fragment registers (M = 1, i.e. max_diffusion_step = 0, has no chain at all)
template <int M, int NKS>
constexpr int poly_slots() { return poly_chains<M, NKS>() > 0 ? poly_chains<M, NKS>() : 1; }

template <int M, int NKS, bool ADJ>
__device__ __forceinline__ void load_poly_frags(const float* Pl, float (&pf)[poly_slots<M, NKS>()][NKS], int lr, int lg) {
    if constexpr (NKS == 5) {
#pragma unroll
        for (int c = 0; c < M - 1; ++c)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                pf[c][ks] = ADJ ? Pl[c * kPFloats + (4 * ks + lg) * kPStride + lr] : Pl[c * kPFloats + lr * kPStride + 4 * ks + lg];
#pragma unroll
        for (int pc = 0; pc < (M - 1 + 3) / 4; ++pc) {
            const int hop = 4 * pc + (lr >> 2), r = 16 + (lr & 3);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                float v = 0.f;
                if (hop < M - 1) v = ADJ ? Pl[hop * kPFloats + (4 * ks + lg) * kPStride + r] : Pl[hop * kPFloats + r * kPStride + 4 * ks + lg];
                pf[M - 1 + pc][ks] = v;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < (M - 1) * 2; ++c) {
            const int m1 = c >> 1, rt = c & 1;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                pf[c][ks] = ADJ ? Pl[m1 * kPFloats + (4 * ks + lg) * kPStride + rt * 16 + lr]
                                : Pl[m1 * kPFloats + (rt * 16 + lr) * kPStride + 4 * ks + lg];
        }
    }
}

// Diffuse ONE 16-column tile: source buf[:, src_col : src_col+16) (rows = nodes) -> slots
// m = 1..M-1 at columns src_col + m*slot_w.  The wave reads the NKS feature fragments once and runs
// all chains (see above) on them.  Issued transposed (features as A operand, polynomial as B operand) so
// that a lane ends up with 4 consecutive columns of one node row: one ds_write_b128 per chain.
// `buf` is an XOR-swizzled tile (common.h lds_sw; stride % 64 == 0, slot_w % 16 == 0).
// Wave-local use: when the source tile was written by this same wave, only EEG_WAVE_SYNC() (no
// workgroup barrier) is needed before the call.
// gout != nullptr: the hop rows are also stored to global planes (forward by-product kept for the
// weight-gradient GEMMs): plane m at gout + (m-1)*gplane, element (node, col) at node*slot_w + col.
// ROWS < 32: the LDS tile holds only ROWS node rows (ROWS >= 4*NKS); result rows beyond are dropped.
// (NKS == 5: rows 20..31 of the hop slots are never written -- nothing reads them in that regime.)
// gbuf (wave-uniform): the global copies leave through a buffer descriptor on `gout` (per-lane 32-bit offsets that do not change over a
// sequence; the step offset is in the scalar base) instead of 64-bit per-lane addresses, which cost two VALU instructions per
// store and step -- VALU time is matrix-pipe time for fp32.  The caller guarantees (M-1) * gplane * 4 < 2^31 (the descriptors span 2 GB; accesses beyond are dropped by the hardware).
template <int M, int NKS, int ROWS = 32>
__device__ __forceinline__ void lds_diffuse_tile(float* buf, int stride, int src_col, int slot_w,
                                                 const float (&pf)[poly_slots<M, NKS>()][NKS], int lr, int lg,
                                                 float* __restrict__ gout = nullptr, size_t gplane = 0, int n_nodes = 0,
                                                 bool gbuf = false) {
    constexpr int NC = poly_chains<M, NKS>();
    if constexpr (NC == 0) return;                       // max_diffusion_step = 0: nothing to mix
    constexpr int NA = NC > 0 ? NC : 1;
    float b[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) b[ks] = buf[lds_sw(4 * ks + lg, src_col + lr, stride)];
    f32x4 acc[NA];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = mfma16(b[ks], pf[c][ks], acc[c]);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int node, hop;                                   // the (hop plane, node row) this lane's 4 columns belong to
        bool live = true;
        if constexpr (NKS == 5) {
            if (c < M - 1) { node = lr; hop = c; }
            else { node = 16 + (lr & 3); hop = 4 * (c - (M - 1)) + (lr >> 2); live = hop < M - 1; }
        } else {
            node = (c & 1) * 16 + lr; hop = c >> 1;
        }
        const float4 v = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
        if (live && (ROWS == 32 || node < ROWS))
            *reinterpret_cast<float4*>(buf + lds_sw(node, (hop + 1) * slot_w + src_col + 4 * lg, stride)) = v;
        if (gout != nullptr && live && node < n_nodes) {
            if (gbuf)
                wbuf_st4(make_wbuf(gout), (unsigned)hop * (unsigned)gplane + (unsigned)(node * slot_w + src_col + 4 * lg), 0u, acc[c]);
            else
                *reinterpret_cast<float4*>(gout + (size_t)hop * gplane + node * slot_w + src_col + 4 * lg) = v;
        }
    }
}

// The same for a PLAIN [rows][stride] tile whose hop slots are slot_w columns wide (the step-input tile of the persistent decoder:
// kernels_decoder.h): source buf[:, src_col : src_col + 16) -> slot m at column m * slot_w + src_col, m = 1..M-1.  gout != nullptr: hop
// plane m also goes to gout + (m-1) * gplane, element (node, col) at node * g_ld + col, for the columns below g_ld (g_ld % 4 == 0).
template <int M, int NKS, int ROWS>
__device__ __forceinline__ void lds_diffuse_tile_plain(float* buf, int stride, int src_col, int slot_w,
                                                       const float (&pf)[poly_slots<M, NKS>()][NKS], int lr, int lg,
                                                       float* __restrict__ gout, size_t gplane, int g_ld, int n_nodes) {
    constexpr int NC = poly_chains<M, NKS>();
    if constexpr (NC == 0) return;
    constexpr int NA = NC > 0 ? NC : 1;
    float b[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) b[ks] = buf[(4 * ks + lg) * stride + src_col + lr];
    f32x4 acc[NA];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = mfma16(b[ks], pf[c][ks], acc[c]);
    const int col = src_col + 4 * lg;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int node, hop;
        bool live = true;
        if constexpr (NKS == 5) {
            if (c < M - 1) { node = lr; hop = c; }
            else { node = 16 + (lr & 3); hop = 4 * (c - (M - 1)) + (lr >> 2); live = hop < M - 1; }
        } else {
            node = (c & 1) * 16 + lr; hop = c >> 1;
        }
        const float4 v = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
        if (live && (ROWS == 32 || node < ROWS)) *reinterpret_cast<float4*>(buf + node * stride + (hop + 1) * slot_w + col) = v;
        if (gout != nullptr && live && node < n_nodes && col < g_ld)
            *reinterpret_cast<float4*>(gout + (size_t)hop * gplane + node * g_ld + col) = v;
    }
}

}  // namespace eeg
