// TEST INFRASTRUCTURE ONLY — a tiny single-OS-thread SIMT emulator used by tests/ to execute the
// HIP kernel SOURCES of eeg_gnn_ssl_amd/csrc on the CPU (no GPU in the build container; GPU
// minutes are scarce).  It is NOT a CPU fallback: the product library never includes this file
// (it is only reachable through tests/emu/platform_emu.h, which only tests/emu/build_emu.py selects as the platform header).
//
// Model: one fiber (ucontext) per HIP thread, run-to-barrier scheduling inside one workgroup at a
// time; __syncthreads() and the wave-collective v_mfma_f32_16x16x4_f32 are rendezvous points.
// MFMA lane layout as documented for gfx950 (cdna_hip_programming.md §3):
//   A: lane l supplies A[i = l&15][k = l>>4];  B: lane l supplies B[k = l>>4][j = l&15];
//   C/D: lane l, reg r holds element (row = 4*(l>>4) + r, col = l&15);  D = C + sum_k A.B as an
//   fmaf chain in k order.
#pragma once
#include <setjmp.h>
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct alignas(16) float4 { float x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

namespace emu {
struct dim3 {
    unsigned x = 1, y = 1, z = 1;
    dim3() {}
    dim3(unsigned a, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
enum State { RUNNABLE, WAIT_BLOCK, WAIT_WAVE, DONE };
struct Fiber {
    ucontext_t ctx;              // first entry only (makecontext); every later switch is _setjmp / _longjmp: swapcontext costs two
    jmp_buf jb;                  // sigprocmask system calls per switch, and a kernel switches fibers millions of times
    bool started = false;
    char* stack = nullptr;
    State st = RUNNABLE;
    unsigned tid = 0;
    float ma = 0, mb = 0;
    f32x4 mc, md;
    bool sync_only = false;
    bool mfma4 = false;          // this rendezvous is a v_mfma_f32_4x4x1_16B_f32
    bool mfma32 = false;         // this rendezvous is a v_mfma_f32_32x32x2_f32 of (ma, mb) on the 16-register accumulator mc16
    float mc16[16] = {0}, md16[16] = {0};
    bool mfma_bf = false;        // this rendezvous is a v_mfma_f32_16x16x32_bf16 of (bfa, bfb): 8 bf16 per lane and operand, as floats
    float bfa[8] = {0}, bfb[8] = {0};
    int shfl_mask = -1;          // >= 0: this rendezvous is a __shfl_xor with that lane mask
    float shfl_val = 0.f;
    int swap_rows = 0;           // 16 / 32: this rendezvous is a v_permlane16_swap / v_permlane32_swap of (swap_a, swap_b)
    float swap_a = 0.f, swap_b = 0.f;
};
struct Ctx {
    dim3 tIdx, bIdx, bDim, gDim;
    char* smem = nullptr;
};
extern Ctx g;
extern Fiber* cur;
extern jmp_buf sched_jb;
inline void yield_to_sched() {
    if (_setjmp(cur->jb) == 0) _longjmp(sched_jb, 1);
}
inline void sync_block() {
    cur->st = WAIT_BLOCK;
    yield_to_sched();
}
inline f32x4 mfma16(float a, float b, f32x4 c) {
    cur->ma = a;
    cur->mb = b;
    cur->mc = c;
    cur->st = WAIT_WAVE;
    yield_to_sched();
    return cur->md;
}
// v_mfma_f32_4x4x1_16B_f32: 16 blocks of 4 lanes; D[lane l][reg r] = c + A(lane 4*(l/4) + r) * B(lane l)
// (layout measured on gfx950 with tools/micro/mfma4_layout.hip)
inline f32x4 mfma4(float a, float b, f32x4 c) {
    cur->ma = a;
    cur->mb = b;
    cur->mc = c;
    cur->mfma4 = true;
    cur->st = WAIT_WAVE;
    yield_to_sched();
    cur->mfma4 = false;
    return cur->md;
}
// v_mfma_f32_32x32x2_f32: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31], D: lane l, register v -> (8*(v>>2) + 4*(l>>5) + (v&3), l & 31)
typedef float f32x16 __attribute__((ext_vector_type(16)));
inline f32x16 mfma32(float a, float b, f32x16 c) {
    cur->ma = a;
    cur->mb = b;
    for (int v = 0; v < 16; ++v) cur->mc16[v] = c[v];
    cur->mfma32 = true;
    cur->st = WAIT_WAVE;
    yield_to_sched();
    cur->mfma32 = false;
    f32x16 d;
    for (int v = 0; v < 16; ++v) d[v] = cur->md16[v];
    return d;
}
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
inline float bf16_bits_to_float(unsigned short h) { unsigned a = (unsigned)h << 16; float x; memcpy(&x, &a, 4); return x; }
// v_mfma_f32_16x16x32_bf16: A[row = lane&15][k = 8*(lane>>4) + i], B[k = 8*(lane>>4) + i][col = lane&15], D[4*(lane>>4) + r][lane&15]
inline f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    for (int i = 0; i < 8; ++i) {
        cur->bfa[i] = bf16_bits_to_float((unsigned short)a[i]);
        cur->bfb[i] = bf16_bits_to_float((unsigned short)b[i]);
    }
    cur->mc = c;
    cur->mfma_bf = true;
    cur->st = WAIT_WAVE;
    yield_to_sched();
    cur->mfma_bf = false;
    return cur->md;
}
// rendezvous of the 64 lanes of a wave: stands for the lockstep execution of real hardware where
// a wave's LDS writes are visible to its own later LDS reads without a workgroup barrier
inline void wave_sync() {
    cur->sync_only = true;
    cur->st = WAIT_WAVE;
    yield_to_sched();
    cur->sync_only = false;
}
// wave-collective butterfly shuffle (all 64 lanes must call it with the same mask)
inline float shfl_xor(float v, int mask) {
    cur->shfl_mask = mask;
    cur->shfl_val = v;
    cur->st = WAIT_WAVE;
    yield_to_sched();
    cur->shfl_mask = -1;
    return cur->shfl_val;
}
// gfx950 v_permlane32_swap_b32 (a's lanes 32..63 <-> b's lanes 0..31) / v_permlane16_swap_b32 (a's odd 16-lane rows <-> b's even rows)
inline void permlane_swap(float& a, float& b, int rows) {
    cur->swap_rows = rows;
    cur->swap_a = a;
    cur->swap_b = b;
    cur->st = WAIT_WAVE;
    yield_to_sched();
    cur->swap_rows = 0;
    a = cur->swap_a;
    b = cur->swap_b;
}
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
}  // namespace emu

#define threadIdx (emu::g.tIdx)
#define blockIdx (emu::g.bIdx)
#define blockDim (emu::g.bDim)
#define gridDim (emu::g.gDim)
using emu::dim3;
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
inline void __syncthreads() { emu::sync_block(); }
inline float __shfl_xor(float v, int mask) { return emu::shfl_xor(v, mask); }

// hip runtime surface used by the host side of the library (device memory == host memory here)
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    memset(p, v, n);
    return 0;
}
inline hipError_t hipMemcpyAsyncD2D(void* d, const void* s, size_t n, hipStream_t) {
    memcpy(d, s, n);
    return 0;
}

#ifdef EEG_SIMT_EMU_IMPL
namespace emu {
Ctx g;
Fiber* cur = nullptr;
jmp_buf sched_jb;
static const std::function<void()>* body_ptr = nullptr;
static void trampoline() {
    (*body_ptr)();
    cur->st = DONE;
    _longjmp(sched_jb, 1);
}
// run fiber f until it yields (or finishes)
static void resume(Fiber& f) {
    if (_setjmp(sched_jb) == 0) {
        if (!f.started) {
            f.started = true;
            setcontext(&f.ctx);          // does not return
        }
        _longjmp(f.jb, 1);
    }
}
// fiber stacks / LDS image: allocated once and reused (a value-initialised vector would clear 128 MB per launch)
static char* g_stacks = nullptr;
static size_t g_stacks_bytes = 0;
static void run_mfma(std::vector<Fiber>& f, unsigned w0) {
    unsigned nsync = 0;
    for (unsigned l = 0; l < 64; ++l) nsync += f[w0 + l].sync_only;
    if (nsync == 64) {
        for (unsigned l = 0; l < 64; ++l) f[w0 + l].st = RUNNABLE;
        return;
    }
    if (nsync != 0) {
        fprintf(stderr, "emu: wave mixes wave_sync() and mfma at one rendezvous\n");
        abort();
    }
    unsigned nshfl = 0;
    for (unsigned l = 0; l < 64; ++l) nshfl += f[w0 + l].shfl_mask >= 0;
    if (nshfl == 64) {
        const int mask = f[w0].shfl_mask;
        float tmp[64];
        for (unsigned l = 0; l < 64; ++l) {
            if (f[w0 + l].shfl_mask != mask) { fprintf(stderr, "emu: divergent shfl masks\n"); abort(); }
            tmp[l] = f[w0 + l].shfl_val;
        }
        for (unsigned l = 0; l < 64; ++l) {
            f[w0 + l].shfl_val = tmp[(l ^ (unsigned)mask) & 63];
            f[w0 + l].st = RUNNABLE;
        }
        return;
    }
    if (nshfl != 0) {
        fprintf(stderr, "emu: wave mixes shfl and mfma at one rendezvous\n");
        abort();
    }
    unsigned nswap = 0;
    for (unsigned l = 0; l < 64; ++l) nswap += f[w0 + l].swap_rows != 0;
    if (nswap == 64) {
        const int rows = f[w0].swap_rows;
        for (unsigned l = 0; l < 64; ++l)
            if (f[w0 + l].swap_rows != rows) { fprintf(stderr, "emu: divergent permlane swaps\n"); abort(); }
        // a's upper half (rows == 32) / odd 16-lane rows (rows == 16) trade places with b's lower half / even rows
        for (unsigned l = 0; l < 64; ++l) {
            const bool a_side = rows == 32 ? l >= 32 : ((l >> 4) & 1) != 0;
            if (a_side) {
                const unsigned partner = rows == 32 ? l - 32 : l - 16;
                const float t = f[w0 + l].swap_a;
                f[w0 + l].swap_a = f[w0 + partner].swap_b;
                f[w0 + partner].swap_b = t;
            }
        }
        for (unsigned l = 0; l < 64; ++l) f[w0 + l].st = RUNNABLE;
        return;
    }
    if (nswap != 0) {
        fprintf(stderr, "emu: wave mixes permlane swaps and mfma at one rendezvous\n");
        abort();
    }
    unsigned nbf = 0;
    for (unsigned l = 0; l < 64; ++l) nbf += f[w0 + l].mfma_bf;
    if (nbf == 64) {
        static float A[16][32], B[32][16];
        for (unsigned l = 0; l < 64; ++l)
            for (int i = 0; i < 8; ++i) {
                A[l & 15][8 * (l >> 4) + i] = f[w0 + l].bfa[i];
                B[8 * (l >> 4) + i][l & 15] = f[w0 + l].bfb[i];
            }
        for (unsigned l = 0; l < 64; ++l) {
            Fiber& x = f[w0 + l];
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                float acc = x.mc[r];
                for (int k = 0; k < 32; ++k) acc = fmaf(A[row][k], B[k][col], acc);     // (products of bf16 pairs are exact in fp32)
                x.md[r] = acc;
            }
            x.st = RUNNABLE;
        }
        return;
    }
    if (nbf != 0) {
        fprintf(stderr, "emu: wave mixes bf16 and fp32 mfma at one rendezvous\n");
        abort();
    }
    unsigned n32 = 0;
    for (unsigned l = 0; l < 64; ++l) n32 += f[w0 + l].mfma32;
    if (n32 == 64) {
        float A[32][2], B[2][32];
        for (unsigned l = 0; l < 64; ++l) {
            A[l & 31][l >> 5] = f[w0 + l].ma;
            B[l >> 5][l & 31] = f[w0 + l].mb;
        }
        for (unsigned l = 0; l < 64; ++l) {
            Fiber& x = f[w0 + l];
            for (int v = 0; v < 16; ++v) {
                const int row = 8 * (v >> 2) + 4 * (l >> 5) + (v & 3), col = l & 31;
                float acc = x.mc16[v];
                for (int k = 0; k < 2; ++k) acc = fmaf(A[row][k], B[k][col], acc);
                x.md16[v] = acc;
            }
            x.st = RUNNABLE;
        }
        return;
    }
    if (n32 != 0) {
        fprintf(stderr, "emu: wave mixes 32x32x2 and other mfma at one rendezvous\n");
        abort();
    }
    unsigned n4 = 0;
    for (unsigned l = 0; l < 64; ++l) n4 += f[w0 + l].mfma4;
    if (n4 == 64) {
        for (unsigned l = 0; l < 64; ++l) {
            Fiber& x = f[w0 + l];
            for (int r = 0; r < 4; ++r) x.md[r] = fmaf(f[w0 + 4 * (l / 4) + r].ma, x.mb, x.mc[r]);
            x.st = RUNNABLE;
        }
        return;
    }
    if (n4 != 0) {
        fprintf(stderr, "emu: wave mixes 4x4x1 and 16x16x4 mfma at one rendezvous\n");
        abort();
    }
    float A[16][4], B[4][16];
    for (unsigned l = 0; l < 64; ++l) {
        A[l & 15][l >> 4] = f[w0 + l].ma;
        B[l >> 4][l & 15] = f[w0 + l].mb;
    }
    for (unsigned l = 0; l < 64; ++l) {
        Fiber& x = f[w0 + l];
        for (int r = 0; r < 4; ++r) {
            int row = 4 * (l >> 4) + r, col = l & 15;
            float acc = x.mc[r];
            for (int k = 0; k < 4; ++k) acc = fmaf(A[row][k], B[k][col], acc);
            x.md[r] = acc;
        }
        x.st = RUNNABLE;
    }
}
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    const unsigned nt = block.x * block.y * block.z;
    if (nt % 64 != 0) {
        fprintf(stderr, "emu: block size %u not a multiple of 64\n", nt);
        abort();
    }
    const size_t STACK = 256 * 1024;
    std::vector<Fiber> fibers(nt);
    if (g_stacks_bytes < (size_t)nt * STACK) {
        free(g_stacks);
        g_stacks_bytes = (size_t)nt * STACK;
        g_stacks = static_cast<char*>(malloc(g_stacks_bytes));
        if (g_stacks == nullptr) { fprintf(stderr, "emu: out of memory for fiber stacks\n"); abort(); }
    }
    char* const stacks = g_stacks;
    std::vector<char> smem(smem_bytes + 64);
    body_ptr = &body;
    g.bDim = block;
    g.gDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                // poison LDS so reads of never-written shared memory are visible as NaNs
                memset(smem.data(), 0xFF, smem.size());
                g.smem = smem.data();
                for (unsigned t = 0; t < nt; ++t) {
                    Fiber& f = fibers[t];
                    f.st = RUNNABLE;
                    f.tid = t;
                    f.started = false;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = stacks + (size_t)t * STACK;
                    f.ctx.uc_stack.ss_size = STACK;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, (void (*)())trampoline, 0);
                }
                unsigned done = 0;
                while (done < nt) {
                    bool progressed = false;
                    for (unsigned t = 0; t < nt; ++t) {
                        Fiber& f = fibers[t];
                        if (f.st != RUNNABLE) continue;
                        cur = &f;
                        g.bIdx = dim3(bx, by, bz);
                        g.tIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                        resume(f);
                        progressed = true;
                        if (f.st == DONE) ++done;
                    }
                    // wave collectives
                    for (unsigned w0 = 0; w0 < nt; w0 += 64) {
                        unsigned waiting = 0;
                        for (unsigned l = 0; l < 64; ++l) waiting += fibers[w0 + l].st == WAIT_WAVE;
                        if (waiting == 64) {
                            run_mfma(fibers, w0);
                            progressed = true;
                        }
                    }
                    // block barrier: every not-finished thread must be waiting at it
                    unsigned wb = 0;
                    for (unsigned t = 0; t < nt; ++t) wb += fibers[t].st == WAIT_BLOCK;
                    if (wb && wb + done == nt) {
                        for (unsigned t = 0; t < nt; ++t)
                            if (fibers[t].st == WAIT_BLOCK) fibers[t].st = RUNNABLE;
                        progressed = true;
                    }
                    if (!progressed) {
                        fprintf(stderr, "emu: deadlock (divergent barrier / partial-wave MFMA) in block %u,%u\n", bx, by);
                        abort();
                    }
                }
            }
}
}  // namespace emu
#endif
