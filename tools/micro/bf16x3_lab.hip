// Gated experiment (NOT the product path, NOT the headline): the hoisted NN GEMM of the DCGRU layer,
// C[R x 192] = A[R x 192] * W[192 x 192] with fp32 operands and fp32 results, computed as a THREE-TERM bf16 split on the bf16
// matrix pipe:  a = a_hi + a_mid + a_lo,  w = w_hi + w_mid + w_lo  (each term a bf16, together 24 mantissa bits), and
//   a * w ~= a_hi w_hi + (a_hi w_mid + a_mid w_hi) + (a_mid w_mid + a_hi w_lo + a_lo w_hi)       (6 of the 9 products)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Prints time and the error against an fp64 host sum next to the true-fp32
// MFMA kernel of the product (gemm_nnr_kernel).  Answers: can 55 % of the step run > 2x faster inside 2e-5?
//   make -C tools/micro bf16x3_lab && ./bf16x3_lab [rounds]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "../../eeg_gnn_ssl_amd/csrc/kernels_gemm_q.h"
#include "../../eeg_gnn_ssl_amd/csrc/nnq_order.h"

using namespace eeg;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));     // 8 bf16 = 4 VGPRs: one A / B fragment of 16x16x32
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// round-to-nearest-even fp32 -> bf16 (as the upper 16 bits), two at a time
__device__ __forceinline__ unsigned lab_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));      // gfx950: two fp32 -> packed bf16, round to nearest even
    return r;
}
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
// x[0..7] -> three fragments (hi, mid, lo)
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    u32x4 H, M, L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned ph = lab_pk_bf16(a, b);
        const float ra = a - bf16_lo(ph), rb = b - bf16_hi(ph);
        const unsigned pm = lab_pk_bf16(ra, rb);
        const float sa = ra - bf16_lo(pm), sb = rb - bf16_hi(pm);
        H[i] = ph; M[i] = pm; L[i] = lab_pk_bf16(sa, sb);
    }
    h = __builtin_bit_cast(bf16x8, H); m = __builtin_bit_cast(bf16x8, M); l = __builtin_bit_cast(bf16x8, L);
}

// One workgroup = 128 rows x 192 columns; wave w: rows 32w .. 32w+31 (two 16-row tiles) x 12 column tiles.
// Wp: [3 terms][K/32 chunks][12 column tiles][64 lanes][8 bf16]: lane l of column tile ct, chunk c holds W[32c + 8(l>>4) + i][16ct + (l&15)].
// The three 12-KB term blocks of a chunk are staged in LDS (two stages) and shared by the four waves.
// TERMS: 6 (above) or 3 (a_hi w_hi + a_hi w_mid + a_mid w_hi: ~16 mantissa bits).
template <int TERMS>
__global__ __launch_bounds__(256, 2) void gemm_nn_bf16x3_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Wp,
                                                                float* __restrict__ C, int R, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // 2 stages x 3 terms x 12 KB
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const int row0 = blockIdx.x * 128 + 32 * w;
    const int nch = K / 32;
    f32x4 acc[2][12];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const size_t term_stride = (size_t)nch * 12 * 64 * 8;                     // bf16 elements per term
    u32x4 pb[9];                                                               // next chunk's 36 KB = 2304 x 16 B over 256 threads, in flight
    auto fetch_b = [&](int c) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int e = tid + 256 * q, t = e / 768, r = e % 768;
            pb[q] = *reinterpret_cast<const u32x4*>(Wp + t * term_stride + ((size_t)c * 768 + r) * 8);
        }
    };
    auto put_b = [&](int st) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int e = tid + 256 * q, t = e / 768, r = e % 768;
            *reinterpret_cast<u32x4*>(smem + (size_t)st * 36864 + (size_t)t * 12288 + (size_t)r * 16) = pb[q];
        }
    };
    float xa[2][8];
    auto load_a = [&](int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r = row0 + 16 * i + lr;
            if (r >= R) r = R - 1;
            const float4 v0 = *reinterpret_cast<const float4*>(A + (size_t)r * K + 32 * c + 8 * lg);
            const float4 v1 = *reinterpret_cast<const float4*>(A + (size_t)r * K + 32 * c + 8 * lg + 4);
            xa[i][0] = v0.x; xa[i][1] = v0.y; xa[i][2] = v0.z; xa[i][3] = v0.w;
            xa[i][4] = v1.x; xa[i][5] = v1.y; xa[i][6] = v1.z; xa[i][7] = v1.w;
        }
    };
    fetch_b(0);
    load_a(0);
    put_b(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int st = c & 1;
        bf16x8 ah[2], am[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) split3(xa[i], ah[i], am[i], al[i]);
        if (c + 1 < nch) { fetch_b(c + 1); load_a(c + 1); }       // in flight during the MFMAs below
        const unsigned char* sb = smem + (size_t)st * 36864 + (size_t)lane * 16;
        // two column tiles at a time and the partial products outermost: consecutive MFMAs then go to four different accumulators
        // (six products into one accumulator back to back would be one dependent chain); smallest terms first
#pragma unroll
        for (int j = 0; j < 12; j += 2) {
            bf16x8 bh[2], bm[2], bl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bh[u] = *reinterpret_cast<const bf16x8*>(sb + (j + u) * 1024);
                bm[u] = *reinterpret_cast<const bf16x8*>(sb + 12288 + (j + u) * 1024);
                bl[u] = *reinterpret_cast<const bf16x8*>(sb + 24576 + (j + u) * 1024);
            }
#define EEG_BF_TERM(B, Aop)                                                                                           \
            _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                             \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
                    acc[i][j + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[u], Aop[i], acc[i][j + u], 0, 0, 0);
            if (TERMS == 6) {
                EEG_BF_TERM(bh, al)
                EEG_BF_TERM(bl, ah)
                EEG_BF_TERM(bm, am)
            }
            EEG_BF_TERM(bh, am)
            EEG_BF_TERM(bm, ah)
            EEG_BF_TERM(bh, ah)
#undef EEG_BF_TERM
        }
        if (c + 1 < nch) put_b(st ^ 1);                           // (stage st^1 was last read in iteration c-1: behind its barrier)
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = row0 + 16 * i + lr;
        if (r < R) {
#pragma unroll
            for (int j = 0; j < 12; ++j) *reinterpret_cast<f32x4*>(C + (size_t)r * 192 + 16 * j + 4 * lg) = acc[i][j];
        }
    }
}

static unsigned short f2bf(float x) {
    unsigned a; memcpy(&a, &x, 4);
    a += 0x7fffu + ((a >> 16) & 1u);
    return (unsigned short)(a >> 16);
}
static float bf2f(unsigned short h) { unsigned a = (unsigned)h << 16; float x; memcpy(&x, &a, 4); return x; }

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 15;
    const int R = 291840, K = 192, O = 192, nch = K / 32;
    std::vector<float> hA((size_t)R * K), hW((size_t)K * O);
    { unsigned s = 12345; for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; } }
    { unsigned s = 777; for (auto& v : hW) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 32768.0f - 1.0f) * 0.25f; } }
    // bf16 term packs of W in fragment order
    std::vector<unsigned short> hP((size_t)3 * nch * 12 * 64 * 8);
    for (int c = 0; c < nch; ++c)
        for (int ct = 0; ct < 12; ++ct)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 8; ++i) {
                    const float x = hW[(size_t)(32 * c + 8 * (l >> 4) + i) * O + 16 * ct + (l & 15)];
                    const unsigned short h = f2bf(x);
                    const float r1 = x - bf2f(h);
                    const unsigned short m = f2bf(r1);
                    const unsigned short lo = f2bf(r1 - bf2f(m));
                    const size_t e = (((size_t)c * 12 + ct) * 64 + l) * 8 + i, ts = (size_t)nch * 12 * 64 * 8;
                    hP[e] = h; hP[ts + e] = m; hP[2 * ts + e] = lo;
                }
    // quad pack for the fp32 product kernel
    const NnqOrder ko = make_nnq_order(1, K);
    std::vector<float> hQ((size_t)ko.nch * 12 * 256);
    for (size_t e = 0; e < hQ.size(); ++e) {
        const int s = e & 3, lane = (e >> 2) & 63, ct = (e >> 8) % 12, c = (e >> 8) / 12;
        const int k = nnq_k_of(ko, c, lane >> 4, s);
        hQ[e] = k < 0 ? 0.f : hW[(size_t)k * O + 16 * ct + (lane & 15)];
    }
    float *A, *C1, *C2, *Q, *bias; unsigned short* P;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&C1, (size_t)R * O * 4)); CK(hipMalloc(&C2, (size_t)R * O * 4));
    CK(hipMalloc(&Q, hQ.size() * 4)); CK(hipMalloc(&P, hP.size() * 2)); CK(hipMalloc(&bias, 192 * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Q, hQ.data(), hQ.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(P, hP.data(), hP.size() * 2, hipMemcpyHostToDevice)); CK(hipMemset(bias, 0, 192 * 4));
    SegPtrs segs{}; segs.p[0] = A;
    auto run_fp32 = [&] {
        const size_t lds = ((size_t)4 * 128 * 16 + 192) * 4;
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nnr_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm_nnr_kernel<4, 2>), dim3(512, 1), dim3(256), lds, 0, segs, 1, K, R, Q, 12, bias, C1, O, O, 0, 0, 0, 0);
    };
    auto run_bf = [&](int terms) {
        const size_t lds = 2 * 36864;
        if (terms == 6) { hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nn_bf16x3_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((gemm_nn_bf16x3_kernel<6>), dim3((R + 127) / 128), dim3(256), lds, 0, A, P, C2, R, K); }
        else { hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nn_bf16x3_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((gemm_nn_bf16x3_kernel<3>), dim3((R + 127) / 128), dim3(256), lds, 0, A, P, C2, R, K); }
    };
    // fp64 reference on sampled rows
    std::vector<int> rows; for (int i = 0; i < 2000; ++i) rows.push_back((int)(((long long)i * 1000003) % R));
    std::vector<double> ref(rows.size() * (size_t)O);
    for (size_t q = 0; q < rows.size(); ++q)
        for (int o = 0; o < O; ++o) { double s = 0; for (int k = 0; k < K; ++k) s += (double)hA[(size_t)rows[q] * K + k] * hW[(size_t)k * O + o]; ref[q * O + o] = s; }
    auto err = [&](float* dC, const char* name) {
        std::vector<float> h((size_t)R * O); CK(hipMemcpy(h.data(), dC, h.size() * 4, hipMemcpyDeviceToHost));
        double maxe = 0, maxv = 0, sum2 = 0;
        for (size_t q = 0; q < rows.size(); ++q) for (int o = 0; o < O; ++o) {
            const double d = std::fabs(h[(size_t)rows[q] * O + o] - ref[q * O + o]); maxe = std::max(maxe, d); sum2 += d * d; maxv = std::max(maxv, std::fabs(ref[q * O + o])); }
        printf("  %-34s max |err| vs fp64 %.3e, rms %.3e (max |value| %.2f -> relative to it %.2e)\n", name, maxe, std::sqrt(sum2 / (rows.size() * O)), maxv, maxe / maxv);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto fn) { std::vector<float> ms; for (int r = 0; r < rounds; ++r) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t); } std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2]; };
    const double fl = 2.0 * R * (double)K * O, bytes = (double)R * (K + O) * 4;
    printf("== NN GEMM R=%d K=%d O=%d (layer-1 x-part): fp32 MFMA vs 3-term bf16 split\n", R, K, O);
    run_fp32(); CK(hipDeviceSynchronize()); err(C1, "fp32 MFMA (gemm_nnr_kernel)");
    run_bf(6); CK(hipDeviceSynchronize()); err(C2, "bf16 split, 6 products");
    run_bf(3); CK(hipDeviceSynchronize()); err(C2, "bf16 split, 3 products");
    const float t32 = time(run_fp32), t6 = time([&] { run_bf(6); }), t3 = time([&] { run_bf(3); });
    printf("  fp32 MFMA                 %.4f ms  %.1f TF/s (fp32-equivalent)  %.2f TB/s of operand + result bytes\n", t32, fl / t32 / 1e9, bytes / t32 / 1e9);
    printf("  bf16 split, 6 products    %.4f ms  %.1f TF/s                     %.2f TB/s   -> %.2fx\n", t6, fl / t6 / 1e9, bytes / t6 / 1e9, t32 / t6);
    printf("  bf16 split, 3 products    %.4f ms  %.1f TF/s                     %.2f TB/s   -> %.2fx\n", t3, fl / t3 / 1e9, bytes / t3 / 1e9, t32 / t3);
    return 0;
}
