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

// the fragment packs (and, with a basis, the per-frequency packs) of n_cells cells in ONE launch; spacks / basis nullable together
int launch_pack_cells(int n_cells, const float* const* Wg, const float* const* bg, const float* const* Wc, const float* const* bc,
                      const int* Fin, int H, int M, float* const* packs, const float* basis, int N, float* const* spacks, hipStream_t st);

// node mixes: to_nodes = 1: X (S,N,F) -> Xh (N,Sp,F) with U^T (pad rows zeroed); 0: Yh (N,Sp,F) -> Y (S,N,F) with U (+ bias).
// The node-major rows are time-major (r = t*B + b); bm = 1: the sample-major side is the batch-major (B, T, N, F) model input
// node_rows: rows per frequency of the node-major side (its group stride; 0 = T*B rounded up to 16); rows [T*B, node_rows) are zeroed
int launch_spec_mix(int to_nodes, const float* in, const float* basis, const float* bias, int N, int T, int B, int F, int bm,
                    float* out, hipStream_t st, const char* tag, int node_rows = 0);

// pad rows [S, Sp) of every frequency of a node-major (N, Sp, F) tensor <- 0 (no launch when Sp == S)
int launch_spec_zero_pad(float* Xh, int N, int S, int F, hipStream_t st);

// grouped NN: C (N*Sp, 16*nct) = A (N*Sp, F) * W_i;  Wq = block 0 of the per-frequency quad packs, wstride floats apart
// bias (16*nct floats, nullable) + gscale (G floats): the tiles of group g start from gscale[g] * bias
// a_gstride: floats between two groups of A (0 = Sp * F, contiguous)
int launch_nng(const float* A, int F, int Sp, int G, const float* Wq, size_t wstride, int nct, float* C, int num_cus,
               hipStream_t st, const char* tag, const float* bias = nullptr, const float* gscale = nullptr, size_t a_gstride = 0);

// grouped TN: partial [G*spg][F][192] of A (G*Sp, F)^T dY (G*Sp, 192)
struct TngPlan { int ok, KT, planar, nkb, spg, rps; };
TngPlan tng_plan(int F, int Sp, int G, int num_cus);
int launch_tng(const TngPlan& p, const float* A, int F, int Sp, int G, const float* dY, float* partial, hipStream_t st, const char* tag,
               size_t a_gstride = 0);
// h-part pair (F = 64): part_g [G*spg][64][128] = Ah^T dY[:, 0:128], part_c [G*spg][64][64] = Arh^T dY[:, 128:192]; one launch
// (ah_gstride: floats between two groups of Ah, 0 = contiguous; Arh is contiguous)
int launch_tng_pair(const TngPlan& p, const float* Ah, const float* Arh, int Sp, int G, const float* dY, float* part_g, float* part_c,
                    hipStream_t st, const char* tag, size_t ah_gstride = 0);
// grouped NN with register-resident weights (kernels_gemm_f.h): C (G, Sp, 192) = A (G, Sp, K) * Wr_i + gscale[i] * bias; Wr = SpecPack::sxr
// block 0, wstride floats apart.  Returns -1 when the width has no instantiation (the caller takes launch_nng), 0 ok, 2 launch error.
// dev knob 20 = 1: never
int launch_nnf(const float* A, size_t a_gstride, int K, int Sp, int G, const float* Wr, size_t wstride, float* C, int num_cus, hipStream_t st,
               const char* tag, const float* bias, const float* gscale);
// input gradient of a spectral layer in one kernel (kernels_gemm_f.h gemm_dxf_kernel): dX (S, N, 64) = U [dYh_i Wt_i^T]_i; Wtq = SpecPack::sxtq
// block 0.  -1: shape not covered (N > 20 or Fin != 64: the caller runs the grouped GEMM + the node mix), 0 ok, 2 launch error
int launch_dxf(const float* dYh, int Sp, int S, int N, int Fin, const float* Wtq, size_t wstride, const float* basis, float* dX,
               hipStream_t st, const char* tag);
// fused weight-gradient GEMM of a 64-unit cell (kernels_gemm_f.h): the x-part and both h-part problems in one pass over dY;
// partials in the layouts above with ONE split count (spg) for the three.  ok = 0: shape not covered (Fin > 128, H != 64)
struct TnfPlan { int ok, fxt, spg, rps; };
TnfPlan tnf_plan(int Fin, int H, int Sp, int G, int num_cus);
int launch_tnf(const TnfPlan& p, const float* Xh, size_t x_gstride, int Fin, const float* Hh, size_t h_gstride, const float* RHh,
               const float* dY, int Sp, int G, float* part_x, float* part_g, float* part_c, hipStream_t st, const char* tag);
// rows [S, Sp) of every group of a (N, Sp, F) node-major tensor <- 0, Sp any row count >= S
int launch_spec_zero_rows(float* Xh, int N, int S, int Sp, int F, hipStream_t st);

}  // namespace eeg
