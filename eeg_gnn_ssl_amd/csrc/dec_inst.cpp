// Instantiations of the persistent decoder forward kernel (kernels_decoder.h), in their own translation unit.
#include "kernels_decoder.h"
#include "prof.h"
#include "seq_launch.h"

namespace eeg {
namespace {
template <int M, int DX>
int launch_one(const DecFwdArgs& a, size_t lds, hipStream_t st) {
    EEG_SET_MAX_LDS((dec_fwd_persist_kernel<64, M, DX>), lds);
    EEG_LAUNCH_P("fwd_persist", (dec_fwd_persist_kernel<64, M, DX>), dim3(a.B < 256 ? a.B : 256), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
template <int M>
int launch_m(int dx, const DecFwdArgs& a, size_t lds, hipStream_t st) {
    switch (dx) {
        case 25: return launch_one<M, 25>(a, lds, st);
        case 16: return launch_one<M, 16>(a, lds, st);
        case 5: return launch_one<M, 5>(a, lds, st);
        default: return launch_one<M, 4>(a, lds, st);
    }
}
}  // namespace

// 0 ok, 1 unsupported M, 2 launch error.  dx = k-steps per weight group of the layer-0 x-part (25, 16, 5 or 4; (Dout/4) % dx == 0)
int launch_dec_fwd_persist(int M, int dx, const DecFwdArgs& a, size_t lds, hipStream_t st) {
    switch (M) {
        case 1: return launch_m<1>(dx, a, lds, st);
        case 2: return launch_m<2>(dx, a, lds, st);
        case 3: return launch_m<3>(dx, a, lds, st);
        case 4: return launch_m<4>(dx, a, lds, st);
        case 5: return launch_m<5>(dx, a, lds, st);
        case 7: return launch_m<7>(dx, a, lds, st);
        default: return 1;
    }
}
}  // namespace eeg
