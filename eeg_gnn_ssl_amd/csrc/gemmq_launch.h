// Host-side launch interface of the round-3 hoisted GEMMs (kernels_gemm_q.h); the instantiations live in gemmq_inst.cpp.
#pragma once
#include "kernels_gemm.h"

namespace eeg {

// ---- NN: C[R x O] = [segments] @ quad pack + bias (gemm_nnr_kernel) ---------------------------------------------------
// Applies to whole 192-column blocks (nct_total % 12 == 0), F % 4 == 0 with at most two tail chunks, 32-bit offsets.
bool nnq_supported(int nseg, int F, int R, int nct_total, int ldc, int O);
// floats of the quad pack of a (nseg * F) x (16 * nct) right-hand side
size_t nnq_pack_floats(int nseg, int F, int nct);
// 0 ok, 2 launch error.  bt*: batch-major segments (0 = time-major).  num_cus: CUs of the device (2 workgroups each)
int launch_nnq(const SegPtrs& segs, int nseg, int F, int R, const float* Bq, int nct_total, const float* bias, float* C,
               int ldc, int O, int btT, int btB, int btN, int num_cus, hipStream_t st, const char* tag);

// ---- TN: partial[split][nseg*F][O] = sum over the rows of a split of A^T dY[:, ycol0 : ycol0 + O] (gemm_tnq_kernel) -------
struct TnqPlan {
    int ok;                 // 0: the shape is not covered (use gemm_tn_dma_kernel / gemm_tn_kernel)
    int KT, OT, planar;     // template instance
    int nkb, nsplit, rps;   // grid (k-blocks, row splits) and rows per split (multiple of 16)
};
// bt: the A segments are batch-major.  Covered: O in {64, 128, 192}, R % 16 == 0, and either 64-wide planes (any count,
// time-major) or O == 192 with any F % 4 == 0 (the x-part of a 64-unit cell)
TnqPlan tnq_plan(int nseg, int F, int R, int O, bool bt, int num_cus);
int launch_tnq(const TnqPlan& p, const SegPtrs& segs, int nseg, int F, int R, const float* dY, int ldy, int ycol0, int O,
               float* partial, int btT, int btB, int btN, hipStream_t st, const char* tag);

// hg + hc of one cell in one launch (kernels_gemm_q.h gemm_tnq_pair_kernel); -1 = the pair is not covered (launch them one by one)
int launch_tnq_pair(const TnqPlan& pg, const TnqPlan& pc, const SegPtrs& sg, const SegPtrs& sc, int nseg, int F, int R, const float* dY, int ldy,
                    int ycol_g, int Og, float* part_g, int ycol_c, int Oc, float* part_c, hipStream_t st, const char* tag);

}  // namespace eeg
