// Instantiations of the persistent decoder backward kernel (kernels_decoder.h), in their own translation unit.
#include "kernels_decoder.h"
#include "prof.h"
#include "seq_launch.h"

namespace eeg {
namespace {
template <int M, int DT>
int launch_one(const DecBwdArgs& a, size_t lds, hipStream_t st) {
    EEG_SET_MAX_LDS((dec_bwd_persist_kernel<64, M, DT>), lds);
    EEG_LAUNCH_P("bwd_persist", (dec_bwd_persist_kernel<64, M, DT>), dim3(a.B < 256 ? a.B : 256), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
template <int M>
int launch_m(int dt, const DecBwdArgs& a, size_t lds, hipStream_t st) {
    return dt == 5 ? launch_one<M, 5>(a, lds, st) : launch_one<M, 4>(a, lds, st);
}
}  // namespace

// 0 ok, 1 unsupported M, 2 launch error.  dt = k-steps per weight group of the projection transpose (5 or 4; (Dout/4) % dt == 0)
int launch_dec_bwd_persist(int M, int dt, const DecBwdArgs& a, size_t lds, hipStream_t st) {
    switch (M) {
        case 1: return launch_m<1>(dt, a, lds, st);
        case 2: return launch_m<2>(dt, a, lds, st);
        case 3: return launch_m<3>(dt, a, lds, st);
        case 4: return launch_m<4>(dt, a, lds, st);
        case 5: return launch_m<5>(dt, a, lds, st);
        case 7: return launch_m<7>(dt, a, lds, st);
        default: return 1;
    }
}
}  // namespace eeg
