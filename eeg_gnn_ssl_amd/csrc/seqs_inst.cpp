// Instantiations of the streamed-weight BPTT kernel (kernels_seq_stream.h), in their own translation unit.
#include "kernels_seq_stream.h"
#include "prof.h"
#include "seq_launch.h"

namespace eeg {
namespace {
constexpr int kStreamGrid = 512;   // two workgroups per CU
template <int M>
int bwd_one(const SeqBwdArgs& a, hipStream_t st) {
    const size_t lds = seq_stream_bwd_lds_floats(M) * sizeof(float);
    if (2 * lds > kMaxLdsBytes) return 3;
    if ((double)a.T * a.B * a.N * 3 * 64 * sizeof(float) >= 2147483648.0) return 3;   // 32-bit buffer offsets
    EEG_SET_MAX_LDS((seq_bwd_stream_kernel<64, M>), lds);
    EEG_LAUNCH_P("seq_bwd", (seq_bwd_stream_kernel<64, M>), dim3(a.B < kStreamGrid ? a.B : kStreamGrid), dim3(256), lds, st, a.Hseq, a.h0, a.Rs, a.Us,
                 a.Cs, a.dHseq, a.d_at_end, a.d_at_len, a.lengths, a.P, a.p_batched, a.b1, a.b2, a.dXW, a.dh0, a.dbias_part,
                 a.T, a.B, a.N, a.act);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
}  // namespace

// 64 units, at most 20 nodes.  0 ok, 1 unsupported M, 2 launch error, 3 two workgroups do not fit the LDS of a CU
int launch_seq_bwd_stream(int M, const SeqBwdArgs& a, hipStream_t st) {
    switch (M) {
        case 1: return bwd_one<1>(a, st);
        case 2: return bwd_one<2>(a, st);
        case 3: return bwd_one<3>(a, st);
        case 4: return bwd_one<4>(a, st);
        case 5: return bwd_one<5>(a, st);
        default: return 1;
    }
}
}  // namespace eeg
