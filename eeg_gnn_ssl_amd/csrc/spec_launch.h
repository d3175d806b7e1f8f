// Host-side launch interface of the spectral form of the hoisted x-part (kernels_spectral.h, kernels_gemm_g.h); the
// instantiations live in spec_inst.cpp.
#pragma once
#include "kernels_gemm.h"

namespace eeg {

// Shapes the spectral path covers: 64 units (192-column pre-activations), Fin % 4 == 0, <= 32 nodes, 32-bit offsets;
// with need_dx additionally Fin == 64 (the input gradient of a layer above the first).
bool spec_supported(int T, int B, int N, int H, int Fin, int M, int need_dx);

size_t spec_pack_floats(int Fin, int H, int M, int N);
int launch_spec_basis(const float* S, int N, float* basis, hipStream_t st);
int launch_spec_pack(const float* Wg, const float* Wc, const float* basis, int Fin, int H, int M, int N, float* spack, hipStream_t st);

// node mixes: to_nodes = 1: X (S,N,F) -> Xh (N,Sp,F) with U^T (pad rows zeroed); 0: Yh (N,Sp,F) -> Y (S,N,F) with U (+ bias).
// bt = 1: the node-major rows are batch-major (r = b*T + t) while the sample-major side is time-major
int launch_spec_mix(int to_nodes, const float* in, const float* basis, const float* bias, int N, int T, int B, int F, int bt,
                    float* out, hipStream_t st, const char* tag);

// pad rows [S, Sp) of every frequency of a node-major (N, Sp, F) tensor <- 0 (no launch when Sp == S)
int launch_spec_zero_pad(float* Xh, int N, int S, int F, hipStream_t st);

// grouped NN: C (N*Sp, 16*nct) = A (N*Sp, F) * W_i;  Wq = block 0 of the per-frequency quad packs, wstride floats apart
int launch_nng(const float* A, int F, int Sp, int G, const float* Wq, size_t wstride, int nct, float* C, int num_cus,
               hipStream_t st, const char* tag);

// grouped TN: partial [G*spg][F][192] of A (G*Sp, F)^T dY (G*Sp, 192)
struct TngPlan { int ok, KT, planar, nkb, spg, rps; };
TngPlan tng_plan(int F, int Sp, int G, int num_cus);
int launch_tng(const TngPlan& p, const float* A, int F, int Sp, int G, const float* dY, float* partial, hipStream_t st, const char* tag);
// h-part pair (F = 64): part_g [G*spg][64][128] = Ah^T dY[:, 0:128], part_c [G*spg][64][64] = Arh^T dY[:, 128:192]; one launch
int launch_tng_pair(const TngPlan& p, const float* Ah, const float* Arh, int Sp, int G, const float* dY, float* part_g, float* part_c,
                    hipStream_t st, const char* tag);

}  // namespace eeg
