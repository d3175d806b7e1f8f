// gfx950 (MI355X, CDNA4) platform layer of the DCRNN kernels: the device intrinsics, launch macros and the few host-side HIP
// calls the kernel sources and api.cpp use by name.  common.h includes EEG_PLATFORM_HEADER, which is THIS file in every product
// build; a build may name another header with the same interface (the test tree does, to run the kernel sources on host memory
// without a GPU).  Nothing in the product refers to such a substitute.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define EEG_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) float name[]
#define EEG_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__)
#define EEG_SET_MAX_LDS(kern, bytes) \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products; D[lane l][reg r] += A(lane 4*(l/4) + r) * B(lane l)
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_32x32x2_f32: A: lane l -> A[i = l & 31][k = l >> 5]; B: lane l -> B[k = l >> 5][j = l & 31]; D: lane l, register v ->
// (row 8 * (v >> 2) + 4 * (l >> 5) + (v & 3), column l & 31).  Same rate as the 16x16x4 form (64 FLOP / cycle and SIMD) with HALF the
// fragment dwords per FLOP; a fragment is 32 consecutive floats of one operand row per half-wave (conflict-free ds_read_b32 on a
// plain row-major image of any row stride).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// bf16 matrix pipe (opt-in three-term split of the hoisted NN GEMMs, kernels_gemm_bf.h): a fragment of v_mfma_f32_16x16x32_bf16 is
// 8 bf16 = 4 VGPRs per lane (A: row lane&15, k = 8*(lane>>4) + i; B: k = 8*(lane>>4) + i, column lane&15; D as the fp32 16x16 tile)
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// two fp32 -> packed bf16 (lo in bits 0..15), round to nearest even (gfx950 v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
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
// This wave's LDS reads have RETURNED (their data is in registers): what an LDS-DMA into the buffer they read must wait for -- the
// DMA's write reaches LDS through the texture path and is not ordered behind the wave's own queued ds_reads.
#define EEG_LDS_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// counted wait on the vector-memory queue (it retires in order), alone or in front of a raw workgroup barrier
#define EEG_VM_WAIT_BARRIER(n) asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory")
#define EEG_VM_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
template <int N> __device__ __forceinline__ void vm_wait_barrier_n() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// EEG_PIN: the value is (re)defined here as far as the optimizer knows (blocks hoisting / sinking of what computes it);
// EEG_USE: the value is needed here (keeps accumulators of ablated code paths alive).  No instructions.
#define EEG_PIN(v) asm volatile("" : "+v"(v))
#define EEG_USE(v) asm volatile("" ::"v"(v))
// the same for a wave-uniform (scalar-register) value: what is computed from it cannot be hoisted above this point
#define EEG_PIN_S(v) asm volatile("" : "+s"(v))
// gfx950 cross-lane register swaps (VOP1, no LDS): v_permlane32_swap_b32 exchanges lanes 32..63 of `a` with lanes 0..31 of `b`;
// v_permlane16_swap_b32 exchanges the odd 16-lane rows of `a` with the even rows of `b` (a.row1 <-> b.row0, a.row3 <-> b.row2).
// Inline asm: ROCm 7.2's __builtin_amdgcn_permlane{16,32}_swap returns the FIRST result twice (checked in the ISA), and the
// hazard recognizer does not look into asm, so the two wait states a VALU write -> permlane read needs are issued here
// (the compiler emits the same s_nop in front of the builtin; nothing is needed behind it).
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ long long cycle_now() { return (long long)__builtin_readcyclecounter(); }
__device__ __forceinline__ long long realtime_now() { return (long long)__builtin_amdgcn_s_memrealtime(); }   // 100 MHz, chip-wide

// LDS-DMA (gfx950 global_load_lds_dwordx4): every lane of the wave copies 16 bytes from its own global
// address to LDS at `lds_wave_base + lane * 16 B` (the LDS side is lane-linear; lds_wave_base must be
// wave-uniform).  Asynchronous: completes with the vector-memory counter (the compiler waits before
// the next barrier).  wave_uniform(): tell the compiler a value derived from threadIdx is wave-uniform.
__device__ __forceinline__ void lds_dma16(float* lds_wave_base, const float* g) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Streamed weight packs (kernels_decoder.h) are read through a buffer descriptor: the address of a load is
// descriptor base (SGPRs) + a per-lane 32-bit offset (ONE VGPR for all loads of a tile) + a wave-uniform offset
// (SGPR / literal).  With flat loads every k-step row further than 4 KB from the previous one needs its own 64-bit VGPR
// address, which the compiler hoists out of the time loop -- hundreds of registers.  Offsets in floats.
typedef __amdgpu_buffer_rsrc_t wbuf_t;
__device__ __forceinline__ wbuf_t make_wbuf(const float* p) {              // p must be wave-uniform
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
// Descriptor over exactly `nbytes` bytes: accesses whose per-lane offset (voff; keep the scalar offset 0 with these) reaches past the
// end are dropped (stores) / return nothing useful (loads) instead of touching memory -- ragged last tiles without per-lane guards.
__device__ __forceinline__ wbuf_t make_wbuf_n(const float* p, unsigned nbytes) {   // p, nbytes must be wave-uniform
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             (int)__builtin_amdgcn_readfirstlane(nbytes), 0x00020000);
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
// the same with a cache-policy operand (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX>
__device__ __forceinline__ void wbuf_st4_aux(wbuf_t b, unsigned voff, unsigned soff, f32x4 v) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), b, 4u * voff, 4u * soff, AUX);
}
__device__ __forceinline__ void wbuf_st2(wbuf_t b, unsigned voff, unsigned soff, float x, float y) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64((u32x2_){__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y)}, b, 4u * voff, 4u * soff, 0);
}
__device__ __forceinline__ void wbuf_st1(wbuf_t b, unsigned voff, unsigned soff, float x) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), b, 4u * voff, 4u * soff, 0);
}
// LDS-DMA through a descriptor (buffer_load_dwordx4 ... lds): 16 bytes per lane to lds_wave_base + lane * 16; BYTE offsets
__device__ __forceinline__ void wbuf_dma16(wbuf_t b, float* lds_wave_base, unsigned voff_bytes, unsigned soff_bytes) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff_bytes, soff_bytes, 0, 0);
}

// device-scope 64-bit fetch-and-add (the clock probe's sums)
__device__ __forceinline__ unsigned long long atomic_add_u64(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }

namespace eeg {
// Fast activations for the recurrent epilogues: v_exp_f32 / v_rcp_f32 (1 ulp each); absolute
// error of sigmoid/tanh ~2e-7, far inside the 1e-4 parity budget (tests assert 2e-5).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// v_log_f32 (log2; absolute error ~1e-7 for arguments in [0.5, 1), no denormal handling): the mantissa part of a split logarithm
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
// a*b + c with the product rounded first (what two separate framework kernels compute)
__device__ __forceinline__ float unfused_mul_add(float a, float b, float c) {
#pragma clang fp contract(off)
    const float p = a * b;
    return p + c;
}

// ---- host side -------------------------------------------------------------------------------------------------------
constexpr int kPlatformIsDevice = 1;
// CUs of the current device (the persistent GEMMs size their grids by it); queried once per process
inline int platform_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}
inline bool platform_copy_floats(float* dst, const float* src, size_t n, hipStream_t st) {
    return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess;
}
// true while launches on `st` are being recorded into a graph instead of executed
inline bool platform_stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
}  // namespace eeg
