// Host-side launch interface of the templated recurrent kernels.  The (H, M) instantiations are
// compiled in separate translation units (seq_inst.cpp, one per H) to keep build times parallel.
#pragma once
#include "common.h"

namespace eeg {

struct SeqFwdArgs {
    // h0 == nullptr: zero initial state; the kernel then clears the (B,N,H) slot in FRONT of Hseq (Hext slot 0)
    const float *XW, *h0, *P;
    int p_batched;
    const float *bhg, *bhc;
    float *Hseq, *Rs, *Us, *Cs, *RHs;
    float *Hpl, *RHpl;      // optional by-product: hop planes P_m h_{t-1} / P_m (r*h_{t-1}), plane m at + (m-1)*plane_stride
    size_t plane_stride;
    int T, B, N, act;
    long long* probe;
    int variant = 0;        // 1: two waves per SIMD (seq_fwd2_kernel) where it exists
    // spectral form (spec_common.h): where the two-wave kernel runs it takes the pre-activations in the eigenbasis, Yh (N, spec_Sp,
    // 3H) node-major with the bias inside, INSTEAD of XW, writes U^T h_slot to Hh (N, spec_SpE, H) (row slot*B + b, slots 0..T) and
    // U^T (r*h_{t-1}) to RHh (N, spec_Sp, H) (both nullable) instead of hop planes, and sets *spec_done = 1; else nothing of this
    const float *spec_U = nullptr, *Yh = nullptr;
    float *Hh = nullptr, *RHh = nullptr;
    int spec_Sp = 0, spec_SpE = 0;
    int* spec_done = nullptr;
};
struct SeqBwdArgs {
    const float *Hseq, *h0, *Rs, *Us, *Cs, *dHseq, *d_at_end, *d_at_len;
    const long long* lengths;
    const float* P;
    int p_batched;
    const float *b1, *b2;
    float *dXW, *dh0, *dbias_part;
    int T, B, N, act;
    long long* probe;
    int variant = 0;        // 1: two waves per SIMD (seq_bwd2_kernel) or two workgroups per CU (seq_bwd_stream_kernel) where they apply
    // spectral form (spec_common.h): where the two-wave kernel runs it writes dYh = U^T dXW (node-major (N, spec_Sp, 3H); row of
    // (t, b) = t*B + b) INSTEAD of dXW and sets *spec_done = 1; elsewhere dXW is written as usual
    const float* spec_U = nullptr;
    float* dYh = nullptr;
    int spec_Sp = 0;
    int* spec_done = nullptr;
};

// return 0 ok, 1 unsupported M for this H, 2 launch error
int launch_seq_fwd_h16(int M, const SeqFwdArgs& a, hipStream_t st);
int launch_seq_fwd_h32(int M, const SeqFwdArgs& a, hipStream_t st);
int launch_seq_fwd_h64(int M, const SeqFwdArgs& a, hipStream_t st);
int launch_seq_bwd_h16(int M, const SeqBwdArgs& a, hipStream_t st);
int launch_seq_bwd_h32(int M, const SeqBwdArgs& a, hipStream_t st);
int launch_seq_bwd_h64(int M, const SeqBwdArgs& a, hipStream_t st);
bool seq_m_supported(int M);
// streamed weights, two workgroups per CU (kernels_seq_stream.h): 64 units, at most 20 nodes; 3 = does not fit
int launch_seq_bwd_stream(int M, const SeqBwdArgs& a, hipStream_t st);
struct DecFwdArgs;
struct DecBwdArgs;
int launch_dec_fwd_persist(int M, const DecFwdArgs& a, size_t lds, hipStream_t st);
int launch_dec_bwd_persist(int M, int dt, const DecBwdArgs& a, size_t lds, hipStream_t st);


}  // namespace eeg
