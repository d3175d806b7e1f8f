// Instantiations of the persistent decoder forward kernel (kernels_decoder.h), in their own translation unit.
#include "kernels_decoder.h"
#include "prof.h"
#include "seq_launch.h"

namespace eeg {
namespace {
template <int M>
int launch_m(const DecFwdArgs& a, size_t lds, hipStream_t st) {
    EEG_SET_MAX_LDS((dec_fwd_persist_kernel<64, M>), lds);
    EEG_LAUNCH_P("fwd_persist", (dec_fwd_persist_kernel<64, M>), dim3(a.B < 256 ? a.B : 256), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
}  // namespace

// 0 ok, 1 unsupported M, 2 launch error
int launch_dec_fwd_persist(int M, const DecFwdArgs& a, size_t lds, hipStream_t st) {
    switch (M) {
        case 1: return launch_m<1>(a, lds, st);
        case 2: return launch_m<2>(a, lds, st);
        case 3: return launch_m<3>(a, lds, st);
        case 4: return launch_m<4>(a, lds, st);
        case 5: return launch_m<5>(a, lds, st);
        case 7: return launch_m<7>(a, lds, st);
        default: return 1;
    }
}
}  // namespace eeg
