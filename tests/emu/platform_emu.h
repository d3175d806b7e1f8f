// TEST INFRASTRUCTURE ONLY: the platform layer of eeg_gnn_ssl_amd/csrc (same names as csrc/platform.h) on the fiber-based SIMT
// emulator of simt_emu.h, so that the kernel SOURCES, the C-ABI orchestration and the Python host layer can be exercised on a
// machine without a GPU.  Selected by tests/emu/build_emu.py through -DEEG_PLATFORM_HEADER; the product build never sees it.
#pragma once
#include "simt_emu.h"

struct f32x2 { float v[2]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
#define EEG_DYN_SMEM(name) float* name = reinterpret_cast<float*>(emu::g.smem)
#define EEG_LAUNCH(kern, grid, block, smem, stream, ...) \
    emu::launch((grid), (block), (smem), [=]() { kern(__VA_ARGS__); })
#define EEG_SET_MAX_LDS(kern, bytes) ((void)0)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return emu::mfma16(a, b, c); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return emu::mfma4(a, b, c); }
using emu::f32x16;
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return emu::mfma32(a, b, c); }
using emu::bf16x8;
using emu::u32x4;
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return emu::mfma_bf16(a, b, c); }
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {                 // round to nearest even, like v_cvt_pk_bf16_f32
    auto rne = [](float x) { unsigned a; memcpy(&a, &x, 4); a += 0x7fffu + ((a >> 16) & 1u); return a >> 16; };
    return rne(lo) | (rne(hi) << 16);
}
__device__ __forceinline__ void permlane32_swap(float& a, float& b) { emu::permlane_swap(a, b, 32); }
__device__ __forceinline__ void permlane16_swap(float& a, float& b) { emu::permlane_swap(a, b, 16); }
#define EEG_SCHED_FENCE() ((void)0)
#define EEG_WAVE_SYNC() emu::wave_sync()
#define EEG_SETPRIO(p) ((void)0)
#define EEG_LDS_BARRIER() __syncthreads()
#define EEG_LDS_WAIT() ((void)0)
#define EEG_VM_WAIT_BARRIER(n) __syncthreads()
#define EEG_VM_WAIT(n) ((void)0)
template <int N> inline void vm_wait_barrier_n() { __syncthreads(); }
template <int N> inline void vm_wait_n() {}
#define EEG_PIN(v) ((void)0)
#define EEG_USE(v) ((void)0)
#define EEG_PIN_S(v) ((void)0)
__device__ __forceinline__ long long cycle_now() { return 0; }
__device__ __forceinline__ long long realtime_now() { return 0; }

__device__ __forceinline__ void lds_dma16(float* lds_wave_base, const float* g) {
    memcpy(lds_wave_base + 4 * (threadIdx.x & 63), g, 16);
}
__device__ __forceinline__ int wave_uniform(int v) { return v; }

struct wbuf_t { const float* p; size_t n = ~(size_t)0; };          // n: bytes covered (make_wbuf_n); accesses past it are dropped
__device__ __forceinline__ wbuf_t make_wbuf(const float* p) { return wbuf_t{p}; }
__device__ __forceinline__ wbuf_t make_wbuf_n(const float* p, unsigned nbytes) { return wbuf_t{p, nbytes}; }
__device__ __forceinline__ float wbuf_ld(wbuf_t b, unsigned voff, unsigned soff) {
    return 4 * ((size_t)voff + soff) + 4 <= b.n ? b.p[(size_t)voff + soff] : 0.f;
}
__device__ __forceinline__ f32x4 wbuf_ld4(wbuf_t b, unsigned voff, unsigned soff) {
    const float* q = b.p + (size_t)voff + soff;
    return f32x4{q[0], q[1], q[2], q[3]};
}
__device__ __forceinline__ void wbuf_st4(wbuf_t b, unsigned voff, unsigned soff, f32x4 v) {
    if (4 * ((size_t)voff + soff) + 16 > b.n) return;
    float* q = const_cast<float*>(b.p) + (size_t)voff + soff;
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
}
template <int AUX>
__device__ __forceinline__ void wbuf_st4_aux(wbuf_t b, unsigned voff, unsigned soff, f32x4 v) { wbuf_st4(b, voff, soff, v); }
__device__ __forceinline__ void wbuf_st2(wbuf_t b, unsigned voff, unsigned soff, float x, float y) {
    float* q = const_cast<float*>(b.p) + (size_t)voff + soff; q[0] = x; q[1] = y;
}
__device__ __forceinline__ void wbuf_st1(wbuf_t b, unsigned voff, unsigned soff, float x) {
    if (4 * ((size_t)voff + soff) + 4 <= b.n) const_cast<float*>(b.p)[(size_t)voff + soff] = x;
}
__device__ __forceinline__ void wbuf_dma16(wbuf_t b, float* lds_wave_base, unsigned voff_bytes, unsigned soff_bytes) {
    if ((size_t)voff_bytes + soff_bytes + 16 <= b.n) memcpy(lds_wave_base + 4 * (threadIdx.x & 63), reinterpret_cast<const char*>(b.p) + voff_bytes + soff_bytes, 16);
    else memset(lds_wave_base + 4 * (threadIdx.x & 63), 0, 16);
}

__device__ __forceinline__ unsigned long long atomic_add_u64(unsigned long long* p, unsigned long long v) {
    const unsigned long long o = *p;      // fibers run one at a time
    *p = o + v;
    return o;
}

namespace eeg {
__device__ __forceinline__ float fast_exp(float x) { return expf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float fast_exp2(float x) { return exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return log2f(x); }
__device__ __forceinline__ float unfused_mul_add(float a, float b, float c) { volatile float p = a * b; return p + c; }

constexpr int kPlatformIsDevice = 0;
inline int platform_num_cus() { return 4; }          // a small grid exercises the same persistent-kernel paths
inline bool platform_copy_floats(float* dst, const float* src, size_t n, hipStream_t) {
    memcpy(dst, src, n * sizeof(float));
    return true;
}
inline bool platform_stream_is_capturing(hipStream_t) { return false; }
}  // namespace eeg
