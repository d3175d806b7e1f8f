// K order of the quad-packed right-hand sides of gemm_nnr_kernel (kernels_gemm_q.h), shared by the pack kernel.
// The 16-deep chunks never straddle a hop plane: first the a = F/16 whole chunks of every plane, then the leftover 16-byte
// pieces of all planes gathered into tail chunks (F = 100, 3 planes: 18 chunks + 1 tail chunk with one zero piece).
#pragma once
#include "common.h"

namespace eeg {

struct NnqOrder { int nseg, F, a, b, nmain, ntail, nch; };
__host__ __device__ inline NnqOrder make_nnq_order(int nseg, int F) {
    NnqOrder o;
    o.nseg = nseg; o.F = F; o.a = F / 16; o.b = (F / 4) % 4;
    o.nmain = nseg * o.a; o.ntail = (nseg * o.b + 3) / 4; o.nch = o.nmain + o.ntail;
    return o;
}
// logical K index (seg*F + f) of element s of 16-byte piece p of chunk c; -1 = zero padding
__host__ __device__ inline int nnq_k_of(const NnqOrder& o, int c, int p, int s) {
    if (c < o.nmain) return (c / o.a) * o.F + (c % o.a) * 16 + 4 * p + s;
    const int tp = (c - o.nmain) * 4 + p;
    if (tp >= o.nseg * o.b) return -1;
    return (tp / o.b) * o.F + o.a * 16 + (tp % o.b) * 4 + s;
}
}  // namespace eeg
