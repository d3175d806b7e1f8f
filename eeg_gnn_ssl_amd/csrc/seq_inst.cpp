// Instantiations of the recurrent kernels for one hidden size (compile with -DEEG_SEQ_H=16|32|64).
#include "kernels_seq.h"
#include "prof.h"
#include "seq_launch.h"

#ifndef EEG_SEQ_H
#error "compile with -DEEG_SEQ_H=<hidden units>"
#endif
#define EEG_CAT2(a, b) a##b
#define EEG_CAT(a, b) EEG_CAT2(a, b)

namespace eeg {
namespace {

constexpr int kSeqMaxGrid = 256;   // one workgroup per CU; larger batches are walked by the resident workgroups

// NKS = k-steps of the node mix: 5 covers N <= 20 (the 19-electrode graph), 8 covers N <= 32.
template <int H, int M, int NKS>
int fwd_nks(const SeqFwdArgs& a, hipStream_t st) {
    const size_t lds = SeqGeom<H, M>::fwd_lds_floats() * sizeof(float);
    if (lds > kMaxLdsBytes) return 3;
#if defined(EEG_DEV)   // cycle-probe instantiations: dev build only
    if constexpr (H == 64 && M == 3 && NKS == 5) {
        if (a.probe != nullptr && a.variant != 1) {
            EEG_SET_MAX_LDS((seq_fwd_kernel<H, M, NKS, true>), lds);
            EEG_LAUNCH_P("seq_fwd", (seq_fwd_kernel<H, M, NKS, true>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(256), lds, st, a.XW, a.h0, a.P,
                         a.p_batched, a.bhg, a.bhc, a.Hseq, a.Rs, a.Us, a.Cs, a.RHs, a.Hpl, a.RHpl, a.plane_stride, a.T, a.B, a.N, a.act, a.probe);
            return hipGetLastError() == hipSuccess ? 0 : 2;
        }
    }
#endif
    if constexpr (H == 64 && NKS == 5 && M <= 3) {   // M >= 4: the r + c weights of a wave no longer fit in 256 registers
        if (a.variant == 1 && (double)a.T * a.B * a.N * H * sizeof(float) < 2147483648.0      // (32-bit buffer offsets of its stores
            && (double)(M - 1) * a.plane_stride * sizeof(float) < 2147483648.0) {                //  and of the hop planes it leaves behind: 2 GB descriptors)
            const size_t lds2 = lds + 16 * 64 * sizeof(float);      // + the update-gate tile U [16][64]
#if defined(EEG_DEV)
            if constexpr (M == 3) {
                if (a.probe != nullptr && a.Yh == nullptr) {
                    EEG_SET_MAX_LDS((seq_fwd2_kernel<H, M, NKS, true>), lds2);
                    EEG_LAUNCH_P("seq_fwd", (seq_fwd2_kernel<H, M, NKS, true>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(512), lds2, st, a.XW, a.h0, a.P,
                                 a.p_batched, a.bhg, a.bhc, a.Hseq, a.Rs, a.Us, a.Cs, a.RHs, a.Hpl, a.RHpl, a.plane_stride, a.T, a.B, a.N, a.act, a.probe,
                                 nullptr, 0, 0);
                    return hipGetLastError() == hipSuccess ? 0 : 2;
                }
            }
#endif
            if constexpr (M >= 2) {
                if (a.Yh != nullptr && a.spec_U != nullptr && a.spec_done != nullptr && (double)a.N * a.spec_Sp * 3 * H * sizeof(float) < 2147483648.0 &&
                    (double)a.N * a.spec_SpE * H * sizeof(float) < 2147483648.0) {
#if defined(EEG_DEV)
                    if constexpr (M == 3) {
                        if (a.probe != nullptr) {
                            const size_t ldsp = lds2 + (3 * 16 * 64 + 4 * 64 + 4 * (SeqGeom<H, M>::KS / 4) * 256) * sizeof(float);
                            EEG_SET_MAX_LDS((seq_fwd2_kernel<H, M, NKS, true, true>), ldsp);
                            EEG_LAUNCH_P("seq_fwd", (seq_fwd2_kernel<H, M, NKS, true, true>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(512), ldsp, st,
                                         a.Yh, a.h0, a.P, a.p_batched, a.bhg, a.bhc, a.Hseq, a.Rs, a.Us, a.Cs, a.RHs, a.Hh, a.RHh, (size_t)0, a.T, a.B, a.N,
                                         a.act, a.probe, a.spec_U, a.spec_Sp, a.spec_SpE);
                            *a.spec_done = 1;
                            return hipGetLastError() == hipSuccess ? 0 : 2;
                        }
                    }
#endif
                    const size_t lds3 = lds2 + (3 * 16 * 64 + 4 * 64 + 4 * (SeqGeom<H, M>::KS / 4) * 256) * sizeof(float);   // + XR, 2 x XC [16][64], XRr [4][64], W1L
                    EEG_SET_MAX_LDS((seq_fwd2_kernel<H, M, NKS, false, true>), lds3);
                    EEG_LAUNCH_P("seq_fwd", (seq_fwd2_kernel<H, M, NKS, false, true>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(512), lds3, st, a.Yh,
                                 a.h0, a.P, a.p_batched, a.bhg, a.bhc, a.Hseq, a.Rs, a.Us, a.Cs, a.RHs, a.Hh, a.RHh, (size_t)0, a.T, a.B, a.N, a.act, a.probe,
                                 a.spec_U, a.spec_Sp, a.spec_SpE);
                    *a.spec_done = 1;
                    return hipGetLastError() == hipSuccess ? 0 : 2;
                }
            }
            EEG_SET_MAX_LDS((seq_fwd2_kernel<H, M, NKS>), lds2);
            EEG_LAUNCH_P("seq_fwd", (seq_fwd2_kernel<H, M, NKS>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(512), lds2, st, a.XW, a.h0, a.P,
                         a.p_batched, a.bhg, a.bhc, a.Hseq, a.Rs, a.Us, a.Cs, a.RHs, a.Hpl, a.RHpl, a.plane_stride, a.T, a.B, a.N, a.act, a.probe,
                         nullptr, 0, 0);
            return hipGetLastError() == hipSuccess ? 0 : 2;
        }
    }
    EEG_SET_MAX_LDS((seq_fwd_kernel<H, M, NKS>), lds);
    EEG_LAUNCH_P("seq_fwd", (seq_fwd_kernel<H, M, NKS>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(256), lds, st, a.XW, a.h0, a.P, a.p_batched,
                 a.bhg, a.bhc, a.Hseq, a.Rs, a.Us, a.Cs, a.RHs, a.Hpl, a.RHpl, a.plane_stride, a.T, a.B, a.N, a.act, a.probe);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
template <int H, int M>
int fwd_one(const SeqFwdArgs& a, hipStream_t st) {
    return a.N <= 20 ? fwd_nks<H, M, 5>(a, st) : fwd_nks<H, M, 8>(a, st);
}
template <int H, int M, int NKS>
int bwd_nks(const SeqBwdArgs& a, hipStream_t st) {
    const size_t lds = SeqGeom<H, M>::bwd_lds_floats(SeqGeom<H, M>::bwd_rows(NKS)) * sizeof(float);
    if (lds > kMaxLdsBytes) return 3;
#if defined(EEG_DEV)
    if constexpr (H == 64 && M == 3 && NKS == 5) {
        if (a.probe != nullptr && a.variant != 1) {
            EEG_SET_MAX_LDS((seq_bwd_kernel<H, M, NKS, true>), lds);
            EEG_LAUNCH_P("seq_bwd", (seq_bwd_kernel<H, M, NKS, true>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(256), lds, st, a.Hseq, a.h0, a.Rs, a.Us,
                         a.Cs, a.dHseq, a.d_at_end, a.d_at_len, a.lengths, a.P, a.p_batched, a.b1, a.b2, a.dXW, a.dh0,
                         a.dbias_part, a.T, a.B, a.N, a.act, a.probe);
            return hipGetLastError() == hipSuccess ? 0 : 2;
        }
    }
#endif
    if constexpr (H == 64 && NKS == 5 && M <= 3) {   // two waves per SIMD: role A holds w1 + half of w2 (M >= 4: > 256 registers)
        if (a.variant == 1 && (double)a.T * a.B * a.N * 3 * H * sizeof(float) < 2147483648.0) {   // (32-bit buffer offsets)
            // tiles + DP [4][20][20] + two coefficient buffers of 4 x (5*256 + 256 + 64) floats (seq_bwd2_kernel)
            const size_t lds2 = ((size_t)(M - 1) * kPFloats + 32 * (SeqGeom<H, M>::KAP + SeqGeom<H, M>::KGP) + 4 * 20 * 20 + 2 * 4 * (5 * 256 + 256 + 64)) * sizeof(float);
#if defined(EEG_DEV)
            if constexpr (M == 3) {
                if (a.probe != nullptr && a.dYh == nullptr) {
                    EEG_SET_MAX_LDS((seq_bwd2_kernel<H, M, NKS, true>), lds2);
                    EEG_LAUNCH_P("seq_bwd", (seq_bwd2_kernel<H, M, NKS, true>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(512), lds2, st, a.Hseq, a.h0, a.Rs, a.Us,
                                 a.Cs, a.dHseq, a.d_at_end, a.d_at_len, a.lengths, a.P, a.p_batched, a.b1, a.b2, a.dXW, a.dh0,
                                 a.dbias_part, a.T, a.B, a.N, a.act, a.probe, nullptr, nullptr, 0);
                    return hipGetLastError() == hipSuccess ? 0 : 2;
                }
            }
#endif
            if constexpr (M >= 2) {
                if (a.dYh != nullptr && a.spec_U != nullptr && a.spec_done != nullptr && (double)a.N * a.spec_Sp * 3 * H * sizeof(float) < 2147483648.0) {
#if defined(EEG_DEV)
                    if constexpr (M == 3) {
                        if (a.probe != nullptr) {
                            EEG_SET_MAX_LDS((seq_bwd2_kernel<H, M, NKS, true, true>), lds2);
                            EEG_LAUNCH_P("seq_bwd", (seq_bwd2_kernel<H, M, NKS, true, true>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(512), lds2, st,
                                         a.Hseq, a.h0, a.Rs, a.Us, a.Cs, a.dHseq, a.d_at_end, a.d_at_len, a.lengths, a.P, a.p_batched, a.b1, a.b2, a.dXW,
                                         a.dh0, a.dbias_part, a.T, a.B, a.N, a.act, a.probe, a.spec_U, a.dYh, a.spec_Sp);
                            *a.spec_done = 1;
                            return hipGetLastError() == hipSuccess ? 0 : 2;
                        }
                    }
#endif
                    EEG_SET_MAX_LDS((seq_bwd2_kernel<H, M, NKS, false, true>), lds2);
                    EEG_LAUNCH_P("seq_bwd", (seq_bwd2_kernel<H, M, NKS, false, true>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(512), lds2, st, a.Hseq,
                                 a.h0, a.Rs, a.Us, a.Cs, a.dHseq, a.d_at_end, a.d_at_len, a.lengths, a.P, a.p_batched, a.b1, a.b2, a.dXW, a.dh0,
                                 a.dbias_part, a.T, a.B, a.N, a.act, a.probe, a.spec_U, a.dYh, a.spec_Sp);
                    *a.spec_done = 1;
                    return hipGetLastError() == hipSuccess ? 0 : 2;
                }
            }
            EEG_SET_MAX_LDS((seq_bwd2_kernel<H, M, NKS>), lds2);
            EEG_LAUNCH_P("seq_bwd", (seq_bwd2_kernel<H, M, NKS>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(512), lds2, st, a.Hseq, a.h0, a.Rs, a.Us, a.Cs,
                         a.dHseq, a.d_at_end, a.d_at_len, a.lengths, a.P, a.p_batched, a.b1, a.b2, a.dXW, a.dh0,
                         a.dbias_part, a.T, a.B, a.N, a.act, a.probe, nullptr, nullptr, 0);
            return hipGetLastError() == hipSuccess ? 0 : 2;
        }
    }
    EEG_SET_MAX_LDS((seq_bwd_kernel<H, M, NKS>), lds);
    EEG_LAUNCH_P("seq_bwd", (seq_bwd_kernel<H, M, NKS>), dim3(a.B < kSeqMaxGrid ? a.B : kSeqMaxGrid), dim3(256), lds, st, a.Hseq, a.h0, a.Rs, a.Us, a.Cs,
                 a.dHseq, a.d_at_end, a.d_at_len, a.lengths, a.P, a.p_batched, a.b1, a.b2, a.dXW, a.dh0,
                 a.dbias_part, a.T, a.B, a.N, a.act, a.probe);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
template <int H, int M>
int bwd_one(const SeqBwdArgs& a, hipStream_t st) {
    return a.N <= 20 ? bwd_nks<H, M, 5>(a, st) : bwd_nks<H, M, 8>(a, st);
}

}  // namespace

int EEG_CAT(launch_seq_fwd_h, EEG_SEQ_H)(int M, const SeqFwdArgs& a, hipStream_t st) {
    switch (M) {
        case 1: return fwd_one<EEG_SEQ_H, 1>(a, st);
        case 2: return fwd_one<EEG_SEQ_H, 2>(a, st);
        case 3: return fwd_one<EEG_SEQ_H, 3>(a, st);
        case 4: return fwd_one<EEG_SEQ_H, 4>(a, st);
        case 5: return fwd_one<EEG_SEQ_H, 5>(a, st);
        case 7: return fwd_one<EEG_SEQ_H, 7>(a, st);
        default: return 1;
    }
}
int EEG_CAT(launch_seq_bwd_h, EEG_SEQ_H)(int M, const SeqBwdArgs& a, hipStream_t st) {
    switch (M) {
        case 1: return bwd_one<EEG_SEQ_H, 1>(a, st);
        case 2: return bwd_one<EEG_SEQ_H, 2>(a, st);
        case 3: return bwd_one<EEG_SEQ_H, 3>(a, st);
        case 4: return bwd_one<EEG_SEQ_H, 4>(a, st);
        case 5: return bwd_one<EEG_SEQ_H, 5>(a, st);
        case 7: return bwd_one<EEG_SEQ_H, 7>(a, st);
        default: return 1;
    }
}

}  // namespace eeg
