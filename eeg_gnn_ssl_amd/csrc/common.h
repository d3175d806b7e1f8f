// Shared device/host helpers for the gfx950 (MI355X, CDNA4) DCRNN kernels.
//
// Everything here is written for wave64 + v_mfma_f32_16x16x4_f32 (exact fp32 MFMA).  Lane maps
// (cdna_hip_programming.md §3):  A: lane l -> A[i=l&15][k=l>>4];  B: lane l -> B[k=l>>4][j=l&15];
// C/D: lane l, reg r -> (row = 4*(l>>4)+r, col = l&15).
//
// EEG_SIMT_EMU is defined ONLY by tests/emu/build_emu.py, which compiles these same sources
// against a fiber-based emulator so the kernel logic can be checked without a GPU.  The product
// build (Makefile / __graft_entry__.build) never defines it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(EEG_SIMT_EMU)
#include "simt_emu.h"
#define EEG_DYN_SMEM(name) float* name = reinterpret_cast<float*>(emu::g.smem)
#define EEG_LAUNCH(kern, grid, block, smem, stream, ...) \
    emu::launch((grid), (block), (smem), [=]() { kern(__VA_ARGS__); })
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return emu::mfma16(a, b, c); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return emu::mfma4(a, b, c); }
#define EEG_SCHED_FENCE() ((void)0)
#define EEG_WAVE_SYNC() emu::wave_sync()
#define EEG_SETPRIO(p) ((void)0)
#define EEG_LDS_BARRIER() __syncthreads()
__device__ __forceinline__ long long cycle_now() { return 0; }
#else
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define EEG_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) float name[]
#define EEG_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products; D[lane l][reg r] += A(lane 4*(l/4) + r) * B(lane l)
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
// pins the instruction order at this point (keeps hand-placed LDS prefetches ahead of the MFMAs)
#define EEG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// A wave executes in lockstep and its LDS operations complete in order, so data a wave wrote to LDS
// is visible to its own later LDS reads; this only stops the compiler from reordering across it.
#define EEG_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#define EEG_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + barrier, and the
// fence makes the compiler drain the vector-memory counter (s_waitcnt vmcnt(0)) in front of every s_barrier: a wave
// with global loads or stores in flight -- the recurrent kernels prefetch their operands a step ahead and stream
// their results out -- then sits at the barrier for a full HBM round trip, every step (measured: ~2200 cycles per
// barrier that follows a prefetch).  The waves of these kernels exchange data through LDS only (no wave reads global
// memory another wave of its workgroup wrote in the same launch), so waiting for the LDS counter is sufficient; the
// compiler still tracks the outstanding loads and waits where their registers are first used.
#define EEG_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__device__ __forceinline__ long long cycle_now() { return (long long)__builtin_readcyclecounter(); }
#endif

// LDS-DMA (gfx950 global_load_lds_dwordx4): every lane of the wave copies 16 bytes from its own global
// address to LDS at `lds_wave_base + lane * 16 B` (the LDS side is lane-linear; lds_wave_base must be
// wave-uniform).  Asynchronous: completes with the vector-memory counter (the compiler waits before
// the next barrier).  wave_uniform(): tell the compiler a value derived from threadIdx is wave-uniform.
#if defined(EEG_SIMT_EMU)
__device__ __forceinline__ void lds_dma16(float* lds_wave_base, const float* g) {
    memcpy(lds_wave_base + 4 * (threadIdx.x & 63), g, 16);
}
__device__ __forceinline__ int wave_uniform(int v) { return v; }
#else
__device__ __forceinline__ void lds_dma16(float* lds_wave_base, const float* g) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// Streamed weight packs (kernels_decoder.h) are read through a buffer descriptor: the address of a load is
// descriptor base (SGPRs) + a per-lane 32-bit offset (ONE VGPR for all loads of a tile) + a wave-uniform offset
// (SGPR / literal).  With flat loads every k-step row further than 4 KB from the previous one needs its own 64-bit VGPR
// address, which the compiler hoists out of the time loop -- hundreds of registers.  Offsets in floats.
#if defined(EEG_SIMT_EMU)
struct wbuf_t { const float* p; };
__device__ __forceinline__ wbuf_t make_wbuf(const float* p) { return wbuf_t{p}; }
__device__ __forceinline__ float wbuf_ld(wbuf_t b, unsigned voff, unsigned soff) { return b.p[(size_t)voff + soff]; }
__device__ __forceinline__ f32x4 wbuf_ld4(wbuf_t b, unsigned voff, unsigned soff) {
    const float* q = b.p + (size_t)voff + soff;
    return f32x4{q[0], q[1], q[2], q[3]};
}
__device__ __forceinline__ void wbuf_st4(wbuf_t b, unsigned voff, unsigned soff, f32x4 v) {
    float* q = const_cast<float*>(b.p) + (size_t)voff + soff;
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
}
#else
typedef __amdgpu_buffer_rsrc_t wbuf_t;
__device__ __forceinline__ wbuf_t make_wbuf(const float* p) {              // p must be wave-uniform
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float wbuf_ld(wbuf_t b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, 4u * voff, 4u * soff, 0));
}
// 16-byte accesses of activations through a descriptor (offsets in floats, < 2^29)
__device__ __forceinline__ f32x4 wbuf_ld4(wbuf_t b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, 4u * voff, 4u * soff, 0));
}
__device__ __forceinline__ void wbuf_st4(wbuf_t b, unsigned voff, unsigned soff, f32x4 v) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), b, 4u * voff, 4u * soff, 0);
}
#endif

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
// per-tile stride of the 4x4x1 remainder hand-over scratch: [4 lane groups][4 nodes][16 cols] with the lane groups
// 80 floats apart (80 = 16 mod 32: the two lane groups of a half-wave write different bank halves)
constexpr int kRemTile = 4 * 80;

// columns of the packs c1 / c2 (kernels_pack.h): hidden units, then input features padded so that the decoder's
// layers (Fin <= 128, 64 units) all have the SAME column-tile count, a literal in kernels_decoder.h
__host__ __device__ constexpr int cell_pack_cx_cols(int Fin, int H) { return H + (Fin <= 128 ? 128 : round_up(Fin, 64)); }

// Quad-permuted K order of the recurrent-kernel weight packs: MFMA number `ks` consumes, on lane
// group g = lane>>4, the logical k index 16*(ks/4) + 4*g + (ks%4), so that one ds_read_b128 of
// A[row][16q + 4g .. +3] feeds four consecutive MFMAs.
__host__ __device__ constexpr int kperm(int ks, int g) { return 16 * (ks / 4) + 4 * g + (ks % 4); }

// Fast activations for the recurrent epilogues: v_exp_f32 / v_rcp_f32 (1 ulp each); absolute
// error of sigmoid/tanh ~2e-7, far inside the 1e-4 parity budget (tests assert 2e-5).
#if defined(EEG_SIMT_EMU)
__device__ __forceinline__ float fast_exp(float x) { return expf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif
// a*b + c with the product rounded first (what two separate framework kernels compute)
#if defined(EEG_SIMT_EMU)
__device__ __forceinline__ float unfused_mul_add(float a, float b, float c) { volatile float p = a * b; return p + c; }
#else
__device__ __forceinline__ float unfused_mul_add(float a, float b, float c) {
#pragma clang fp contract(off)
    const float p = a * b;
    return p + c;
}
#endif
__device__ __forceinline__ float sigmoidf_(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * fast_rcp(1.0f + fast_exp(2.0f * x)); }

}  // namespace eeg
