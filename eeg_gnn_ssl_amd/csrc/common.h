// Shared device/host helpers for the gfx950 (MI355X, CDNA4) DCRNN kernels.
//
// Everything here is written for wave64 + v_mfma_f32_16x16x4_f32 (exact fp32 MFMA).  Lane maps
// (cdna_hip_programming.md §3):  A: lane l -> A[i=l&15][k=l>>4];  B: lane l -> B[k=l>>4][j=l&15];
// C/D: lane l, reg r -> (row = 4*(l>>4)+r, col = l&15).
//
// Intrinsics, launch macros and buffer-descriptor accessors come from the platform header (platform.h: gfx950; a build may
// name another header with the same interface through EEG_PLATFORM_HEADER -- the test tree does, for its SIMT emulator).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#ifndef EEG_PLATFORM_HEADER
#define EEG_PLATFORM_HEADER "platform.h"
#endif
#include EEG_PLATFORM_HEADER

namespace eeg {

constexpr int kWave = 64;
constexpr int kMaxNodes = 32;   // node rows are padded to two 16-row MFMA tiles
constexpr size_t kMaxLdsBytes = 160 * 1024;   // LDS of one gfx950 CU
constexpr int kMaxM = 8;        // hop matrices incl. identity (K<=3 with two supports -> 7)

__host__ __device__ constexpr int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ constexpr int round_up(int a, int b) { return ceil_div(a, b) * b; }

// LDS row stride (floats) for an MFMA A-operand tile with K logical columns: K rounded so that
// stride % 32 == 2 -> the 16 rows of a tile land on 16 distinct even banks and the two k-lanes
// (l>>4 = 0/1 in a 32-lane ds_read_b32 group) on even/odd banks: conflict-free fragment reads.
__host__ __device__ constexpr int lds_stride(int k) { return k + ((2 - (k % 32)) + 32) % 32; }

// LDS row stride for MFMA A-operand tiles read with ds_read_b128 (4 consecutive k per lane):
// stride % 64 == 4 -> the 16 rows x 4 dwords of a lane group tile the 64 banks (one residual
// 2-way overlap per group); rows stay 16-byte aligned.
__host__ __device__ constexpr int lds_stride_q(int k) { return k + ((4 - (k % 64)) + 64) % 64; }

// ---- XOR-swizzled LDS tiles of the recurrent kernels ---------------------------------------------------------
// The node-row tiles (rows = nodes, columns = the M hop slots of a feature block) are accessed four ways: b128
// fragment reads (lane (lr, lg): row lr, 16-byte piece 4q + lg), b128 epilogue reads / writes (row lr, piece 4ct + lg),
// b32 node-mix reads (row 4ks + lg, column c0 + lr) and the 4-row remainder reads (row 16 + (lane & 3)).  gfx950
// services ds_read_b128 in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... :
// MI355X_MICROARCH.md §LDS), each holding all 16 values of lr but TWO values of lg, so any layout whose bank slot
// is (f(row) + piece) has a 2-way overlap in every group (round 1: 7-10 % of the kernels' cycles).  Here the row
// stride is a multiple of 64 dwords (every row starts at bank 0) and piece p of row r lives at piece
// (p & ~15) | ((p ^ sigma4(r)) & 15), with sigma4 XOR-linear: sigma4(1) = 4, (2) = 2, (4) = 9, (8) = 8.  Then
//   * b128 reads: slot = sigma4(lr) ^ lg ^ 4q; sigma4(a) ^ sigma4(b) = 1 only for a ^ b = 12, which never pairs two
//     rows of one lane group (nor two of the remainder rows 16..19) -> all four access kinds conflict-free;
//   * b128 writes (8 contiguous lanes, 32 banks): sigma4(0..7) mod 8 are distinct -> conflict-free;
//   * b32 node-mix reads (rows 4ks, 4ks+1 in one half-wave): sigma4(1) = 4 moves the second row to the other 16 banks.
__host__ __device__ constexpr int lds_stride_x(int k) { return round_up(k, 64); }
__host__ __device__ constexpr int sigma4(int r) { return ((r & 1) << 2) ^ (r & 2) ^ ((r & 4) ? 9 : 0) ^ (r & 8); }
// float offset of element (row, col) of a swizzled tile with row stride `stride` (a multiple of 64)
__host__ __device__ constexpr int lds_sw(int row, int col, int stride) {
    return row * stride + ((((col >> 2) & ~15) | (((col >> 2) ^ sigma4(row)) & 15)) << 2) + (col & 3);
}

// Reduce-scatter of a 4x4x1 remainder chain inside the wave, no LDS: register r of lane (lr, lg) holds lane group lg's partial of
// out[node 16 + r][col lr]; the return value of lane (lr, lg) is the complete out[node 16 + lg][col lr] -- every lane ends up with
// ONE element of the 4-node remainder tile (the "one value per lane" layout of the remainder epilogues).  Three register swaps
// (gfx950 v_permlane32_swap / v_permlane16_swap) and three adds; sum order per element: (g0 + g2) + (g1 + g3).
__device__ __forceinline__ float rem4_reduce(f32x4 t) {
    float a = t[0], b = t[1], c = t[2], d = t[3];
    permlane32_swap(a, c);            // a = rows [t0.g0, t0.g1, t2.g0, t2.g1], c = [t0.g2, t0.g3, t2.g2, t2.g3]
    permlane32_swap(b, d);            // b = rows [t1.g0, t1.g1, t3.g0, t3.g1], d = [t1.g2, t1.g3, t3.g2, t3.g3]
    float s02 = a + c, s13 = b + d;   // rows [t0(g0+g2), t0(g1+g3), t2(g0+g2), t2(g1+g3)] / the same of t1, t3
    permlane16_swap(s02, s13);        // s02 = rows [t0, t1, t2, t3](g0+g2), s13 = [t0, t1, t2, t3](g1+g3)
    return s02 + s13;
}

// columns of the packs c1 / c2 (kernels_pack.h): hidden units, then input features padded so that the decoder's
// layers (Fin <= 128, 64 units) all have the SAME column-tile count, a literal in kernels_decoder.h
__host__ __device__ constexpr int cell_pack_cx_cols(int Fin, int H) { return H + (Fin <= 128 ? 128 : round_up(Fin, 64)); }

// Quad-permuted K order of the recurrent-kernel weight packs: MFMA number `ks` consumes, on lane
// group g = lane>>4, the logical k index 16*(ks/4) + 4*g + (ks%4), so that one ds_read_b128 of
// A[row][16q + 4g .. +3] feeds four consecutive MFMAs.
__host__ __device__ constexpr int kperm(int ks, int g) { return 16 * (ks / 4) + 4 * g + (ks % 4); }

// sigmoid / tanh on the platform's fast exp and reciprocal (platform.h: v_exp_f32 / v_rcp_f32)
__device__ __forceinline__ float sigmoidf_(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * fast_rcp(1.0f + fast_exp(2.0f * x)); }
// Four at a time: the scale, the +1 and the final multiply-add as packed fp32 instructions (two per float4), only v_exp_f32 /
// v_rcp_f32 per element -- VALU instructions cost matrix-pipe time in the fp32 recurrent kernels.
__device__ __forceinline__ f32x4 sigmoid4_(f32x4 x) {
    const f32x4 z = x * -1.44269504088896340736f;
    f32x4 d = {fast_exp2(z[0]), fast_exp2(z[1]), fast_exp2(z[2]), fast_exp2(z[3])};
    d = d + 1.0f;
    return (f32x4){fast_rcp(d[0]), fast_rcp(d[1]), fast_rcp(d[2]), fast_rcp(d[3])};
}
__device__ __forceinline__ f32x4 tanh4_(f32x4 x) {
    const f32x4 z = x * (2.0f * 1.44269504088896340736f);
    f32x4 d = {fast_exp2(z[0]), fast_exp2(z[1]), fast_exp2(z[2]), fast_exp2(z[3])};
    d = d + 1.0f;
    const f32x4 r = {fast_rcp(d[0]), fast_rcp(d[1]), fast_rcp(d[2]), fast_rcp(d[3])};
    return 1.0f - 2.0f * r;
}
__device__ __forceinline__ f32x4 relu4_(f32x4 x) {
    return (f32x4){fmaxf(x[0], 0.f), fmaxf(x[1], 0.f), fmaxf(x[2], 0.f), fmaxf(x[3], 0.f)};
}

// ---- dropout (nn.Dropout in front of the classification head, model.py:267, and of the decoder's projection, model.py:191) ------
// Keep decisions come from Philox4x32-10 (Salmon et al., SC'11 -- the counter-based generator torch's own GPU dropout uses): group
// g of four consecutive elements of the dropped tensor takes the four 32-bit words of counter (offset + g) under key `seed`;
// element 4g + j is kept when word j >= thr = p * 2^32 and then scaled by 1 / (1 - p).  Nothing is stored: the backward kernels
// recompute the words from the (seed, offset) pair the forward call used.  The generator state lives in DEVICE memory
// (uint64 {seed, offset}): a forward entry point first launches rng_take_kernel, which copies the pair to `used` and advances
// the offset by the counters the call draws -- so a captured HIP graph draws fresh masks on every replay.
struct DropCfg {
    float p, scale;             // scale = 1 / (1 - p) (0 when p >= 1: everything dropped, like torch)
    unsigned thr;               // keep iff word >= thr
    bool on;                    // p > 0
};
inline DropCfg make_drop_cfg(float p) {
    DropCfg c;
    c.p = p;
    c.on = p > 0.f;
    c.scale = p >= 1.f ? 0.f : 1.f / (1.f - p);
    const double t = (double)p * 4294967296.0;
    c.thr = p >= 1.f ? 0xffffffffu : (unsigned)(t > 4294967295.0 ? 4294967295.0 : t);
    return c;
}
__host__ __device__ inline void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned)p1;
        c3 = (unsigned)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// keep-mask x scale of elements 4g .. 4g+3 (p >= 1: keep nothing -- word >= 0xffffffff would still keep one in 2^32)
__host__ __device__ inline f32x4 dropout_mask4(unsigned long long seed, unsigned long long offset, unsigned long long g,
                                               unsigned thr, float scale) {
    const unsigned long long c = offset + g;
    unsigned w[4];
    philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
    f32x4 m;
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = (scale != 0.f && w[j] >= thr) ? scale : 0.f;
    return m;
}
// the same for ONE element e (the one-value-per-lane remainder epilogues)
__host__ __device__ inline float dropout_mask1(unsigned long long seed, unsigned long long offset, unsigned long long e,
                                               unsigned thr, float scale) {
    const unsigned long long c = offset + (e >> 2);
    unsigned w[4];
    philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
    const unsigned k = (unsigned)e & 3u;
    const unsigned word = k == 0 ? w[0] : (k == 1 ? w[1] : (k == 2 ? w[2] : w[3]));
    return (scale != 0.f && word >= thr) ? scale : 0.f;
}
// 16-byte global accesses of four consecutive floats
__device__ __forceinline__ f32x4 ld4(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return (f32x4){v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

}  // namespace eeg
