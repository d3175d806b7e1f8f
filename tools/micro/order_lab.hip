// Are vector-memory operations of DIFFERENT encodings retired in issue order on gfx950?  The counted waits of the round-3
// GEMMs assume vmcnt is an in-order queue.  Test: G = global_load_lds_dwordx4 from a cold (HBM) address, then B = a
// cache-hot buffer_load_dwordx4 into a VGPR, then s_waitcnt vmcnt(1) (only the YOUNGER op may be outstanding) and a read of
// the LDS bytes G was to deliver.  A stale read = G was overtaken.  Also the buffer-encoded LDS-DMA as the older op.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t wbuf_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int KIND>   // 0: older = global_load_lds (FLAT encoding); 1: older = buffer_load ... lds (MUBUF)
__global__ __launch_bounds__(256) void order_kernel(const float* __restrict__ cold, const float* __restrict__ hot, int iters, unsigned* __restrict__ bad) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const wbuf_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)hot, 0, 0x7fffffff, 0x00020000);
    const wbuf_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cold, 0, 0x7fffffff, 0x00020000);
    float* mine = sm + w * 256;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        // poison the LDS slot, make sure the poison is there
        *reinterpret_cast<f32x4*>(mine + 4 * lane) = (f32x4){-1.f, -1.f, -1.f, -1.f};
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
        const size_t idx = ((size_t)(blockIdx.x * 4 + w) * iters + it) * 4096 + 4 * lane;     // 16 KB apart: always a fresh line
        if (KIND == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cold + idx), (__attribute__((address_space(3))) void*)mine, 16, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (__attribute__((address_space(3))) void*)mine, 16, (unsigned)(4 * lane) * 4u, (unsigned)(idx - 4 * lane) * 4u, 0, 0);
        f32x4 h;
        const unsigned vo = (unsigned)lane * 16u, so = 0u;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(h) : "v"(vo), "s"(rh), "s"(so) : "memory");
        asm volatile("s_waitcnt vmcnt(1)" : "+v"(h) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 got = *reinterpret_cast<volatile f32x4*>(mine + 4 * lane);
        if (got[0] != cold[idx]) ++nbad;               // (cold holds its own index pattern, never -1)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(h) : : "memory");
        if (h[0] == 12345.f) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

// Round 6: may a younger STORE be retired before an older load?  older = buffer_load ... lds from a cold line, then NST buffer stores of
// 16 bytes per lane to this wave's scratch rows (the 16-row x 64-byte pattern of the GEMM epilogue), then s_waitcnt vmcnt(NST): only
// the stores may still be outstanding if the queue retires in order.  A stale LDS read = the load was overtaken by the stores.
template <int NST>
__global__ __launch_bounds__(256) void order_store_kernel(const float* __restrict__ cold, float* __restrict__ scratch, int iters, unsigned* __restrict__ bad) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const wbuf_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cold, 0, 0x7fffffff, 0x00020000);
    const wbuf_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)scratch, 0, 0x7fffffff, 0x00020000);
    float* mine = sm + w * 256;
    unsigned nbad = 0;
    const unsigned srow = (unsigned)((blockIdx.x * 4 + w) * 16 + (lane & 15)) * 768u + (unsigned)(lane >> 4) * 16u;     // bytes: 192-float rows
    for (int it = 0; it < iters; ++it) {
        *reinterpret_cast<f32x4*>(mine + 4 * lane) = (f32x4){-1.f, -1.f, -1.f, -1.f};
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
        const size_t idx = ((size_t)(blockIdx.x * 4 + w) * iters + it) * 4096 + 4 * lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (__attribute__((address_space(3))) void*)mine, 16, (unsigned)(4 * lane) * 4u, (unsigned)(idx - 4 * lane) * 4u, 0, 0);
        const f32x4 v = {(float)it, 1.f, 2.f, 3.f};
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const unsigned so = (unsigned)(k % 12) * 64u;
            asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(v), "v"(srow), "s"(rs), "s"(so) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 got = *reinterpret_cast<volatile f32x4*>(mine + 4 * lane);
        if (got[0] != cold[idx]) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    const int G = 256, iters = 100;     // 1.7 GB of cold lines (32-bit buffer offsets)
    const size_t n = (size_t)G * 4 * iters * 4096 + 4096;
    float *cold, *hot; unsigned* bad;
    CK(hipMalloc(&cold, n * 4)); CK(hipMalloc(&hot, 4096)); CK(hipMalloc(&bad, 4));
    std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 1000003) + 1.f;
    CK(hipMemcpy(cold, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(hot, 0, 4096));
    for (int kind = 0; kind < 2; ++kind) {
        CK(hipMemset(bad, 0, 4));
        if (kind == 0) hipLaunchKernelGGL(order_kernel<0>, dim3(G), dim3(256), 4096, 0, cold, hot, iters, bad);
        else hipLaunchKernelGGL(order_kernel<1>, dim3(G), dim3(256), 4096, 0, cold, hot, iters, bad);
        CK(hipDeviceSynchronize());
        unsigned b; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
        printf("older op = %s, younger = buffer_load to VGPR, s_waitcnt vmcnt(1): %u stale lane-reads of %zu\n",
               kind == 0 ? "global_load_lds (FLAT/global encoding)" : "buffer_load ... lds (MUBUF)", b, (size_t)G * 256 * iters);
    }
    float* scratch; CK(hipMalloc(&scratch, (size_t)G * 4 * 16 * 768));
    for (int nst = 0; nst < 2; ++nst) {
        CK(hipMemset(bad, 0, 4));
        if (nst == 0) hipLaunchKernelGGL(order_store_kernel<3>, dim3(G), dim3(256), 4096, 0, cold, scratch, iters, bad);
        else hipLaunchKernelGGL(order_store_kernel<24>, dim3(G), dim3(256), 4096, 0, cold, scratch, iters, bad);
        CK(hipDeviceSynchronize());
        unsigned b; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
        printf("older op = buffer_load ... lds (cold), younger = %d buffer stores, s_waitcnt vmcnt(%d): %u stale lane-reads of %zu\n",
               nst == 0 ? 3 : 24, nst == 0 ? 3 : 24, b, (size_t)G * 256 * iters);
    }
    return 0;
}
