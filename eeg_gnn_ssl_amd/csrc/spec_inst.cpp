// Instantiations + launch logic of the spectral form of the hoisted x-part, in their own translation unit.
#include "kernels_gemm_g.h"
#include "kernels_gemm_f.h"
#include "kernels_spectral.h"
#include "pack_cell.h"
#include "spec_launch.h"
#include "prof.h"

namespace eeg {

// The packs of ALL cells of an encoder in one launch (their weights change with every optimisation step): blockIdx.y = job
// (a cell's fragment packs, kernels_pack.h, or its per-frequency packs above), blockIdx.x strides over the job's elements.
constexpr int kMaxPackJobs = 8;
struct PackJob {
    const float *Wg, *bg, *Wc, *bc, *basis;
    float* out;
    int spectral;          // 0: pack_cell_body (cp), 1: pack_spectral_body (sp)
    CellPack cp;
    SpecPack sp;
};
struct PackJobs { PackJob j[kMaxPackJobs]; };
__global__ void pack_cells_kernel(PackJobs jobs) {
    const PackJob& jb = jobs.j[blockIdx.y];
    if (jb.spectral) pack_spectral_body(jb.Wg, jb.Wc, jb.basis, jb.out, jb.sp, (int)blockIdx.x, (int)gridDim.x);
    else pack_cell_body(jb.Wg, jb.bg, jb.Wc, jb.bc, jb.out, jb.cp, (int)blockIdx.x, (int)gridDim.x);
}


bool spec_supported(int T, int B, int N, int H, int Fin, int M, int need_dx) {
    if (T < 1 || B < 1 || N < 2 || N > kMaxNodes || H != 64 || Fin < 4 || Fin % 4 != 0 || M < 2 || M > kMaxM) return false;
    if (need_dx && Fin != 64) return false;
    if (Fin / 4 > 256 || make_nnq_order(1, Fin).ntail > 1) return false;
    const double rows = (double)N * spec_rows(T * B);
    // 2 GB buffer descriptors on every operand (platform.h make_wbuf) and 32-bit float4 indices in the mixes
    return rows * (Fin > 192 ? Fin : 192) * 4.0 < 2147483648.0;
}

size_t spec_pack_floats(int Fin, int H, int M, int N) { return make_spec_pack(Fin, H, M, N).total; }

int launch_spec_basis(const float* S, int N, float* basis, hipStream_t st) {
    EEG_SET_MAX_LDS(spectral_basis_kernel, kSpecBasisLds);
    EEG_LAUNCH_P("spec_basis", spectral_basis_kernel, dim3(1), dim3(256), kSpecBasisLds, st, S, N, basis);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_spec_pack(const float* Wg, const float* Wc, const float* basis, int Fin, int H, int M, int N, float* spack, hipStream_t st) {
    const SpecPack p = make_spec_pack(Fin, H, M, N);
    EEG_LAUNCH_P("pack_cell", pack_spectral_kernel, dim3(1024), dim3(256), 0, st, Wg, Wc, basis, spack, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_pack_cells(int n_cells, const float* const* Wg, const float* const* bg, const float* const* Wc, const float* const* bc,
                      const int* Fin, int H, int M, float* const* packs, const float* basis, int N, float* const* spacks, hipStream_t st) {
    PackJobs jobs{};
    int n = 0;
    for (int c = 0; c < n_cells; ++c) {
        if (n + (spacks != nullptr ? 2 : 1) > kMaxPackJobs) return 1;
        PackJob& a = jobs.j[n++];
        a.Wg = Wg[c]; a.bg = bg[c]; a.Wc = Wc[c]; a.bc = bc[c]; a.basis = nullptr; a.out = packs[c]; a.spectral = 0;
        a.cp = make_cell_pack(Fin[c], H, M);
        if (spacks != nullptr) {
            PackJob& b = jobs.j[n++];
            b.Wg = Wg[c]; b.bg = nullptr; b.Wc = Wc[c]; b.bc = nullptr; b.basis = basis; b.out = spacks[c]; b.spectral = 1;
            b.sp = make_spec_pack(Fin[c], H, M, N);
        }
    }
    EEG_LAUNCH_P("pack_cell", pack_cells_kernel, dim3(256, n), dim3(256), 0, st, jobs);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_spec_mix(int to_nodes, const float* in, const float* basis, const float* bias, int N, int T, int B, int F, int bm,
                    float* out, hipStream_t st, const char* tag, int node_rows) {
    const int S = T * B, Sp = node_rows > 0 ? node_rows : spec_rows(S), F4 = F / 4;
#ifndef EEG_X_MIX_VALU
    // the mixes on the matrix pipe where a row is whole 128-byte tiles (measured at cfg2: F = 64 from nodes 0.046 -> 0.037 ms; F = 100
    // to nodes 0.045 -> 0.070: a fourth tile with 4 of 32 columns and rows that straddle lines -- that one keeps the VALU form)
    if (N <= 32 && F % 32 == 0 && (double)(to_nodes ? Sp : S) * (F / 32) < 2.0e9) {
        const int units = (to_nodes ? Sp : S) * ceil_div(F, 32);
        int nb = ceil_div(units, 4);
        if (nb > 2048) nb = 2048;
        if (N <= 20) {
            if (to_nodes) EEG_LAUNCH_P(tag, (spec_mix_mfma_kernel<0, 10>), dim3(nb), dim3(256), 0, st, in, basis, bias, N, S, Sp, F, bm, T, B, out);
            else EEG_LAUNCH_P(tag, (spec_mix_mfma_kernel<1, 10>), dim3(nb), dim3(256), 0, st, in, basis, bias, N, S, Sp, F, bm, T, B, out);
        } else {
            if (to_nodes) EEG_LAUNCH_P(tag, (spec_mix_mfma_kernel<0, 16>), dim3(nb), dim3(256), 0, st, in, basis, bias, N, S, Sp, F, bm, T, B, out);
            else EEG_LAUNCH_P(tag, (spec_mix_mfma_kernel<1, 16>), dim3(nb), dim3(256), 0, st, in, basis, bias, N, S, Sp, F, bm, T, B, out);
        }
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
#endif
    if (N == 19 && F4 <= 128) {
        int threads = 256;
        while (threads > 64 && (threads / 2) >= F4 && (threads / 2) / F4 >= Sp) threads /= 2;
        const int SPW = threads / F4;
        int nb = ceil_div(to_nodes ? Sp : S, SPW);
        if (nb > 1024) nb = 1024;                          // ~4 workgroups per CU, each walking consecutive passes (cf. diffuse_fwd)
        if (to_nodes) EEG_LAUNCH_P(tag, spec_mix_in_kernel<19>, dim3(nb), dim3(threads), 0, st, in, basis, S, Sp, F, bm, T, B, out);
        else EEG_LAUNCH_P(tag, spec_mix_out_kernel<19>, dim3(nb), dim3(threads), 0, st, in, basis, bias, S, Sp, F, bm, T, B, out);
    } else {
        const size_t total = (size_t)(to_nodes ? Sp : S) * N * F4;
        int nb = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        EEG_LAUNCH_P(tag, spec_mix_generic_kernel, dim3(nb), dim3(256), (size_t)N * N * sizeof(float), st, in, basis, bias, N, S, Sp, F, bm,
                     T, B, to_nodes, out);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_spec_zero_rows(float* Xh, int N, int S, int Sp, int F, hipStream_t st) {
    if (Sp <= S) return 0;
    EEG_LAUNCH_P("zero", spec_zero_pad_kernel, dim3(ceil_div(N * (Sp - S) * F, 256)), dim3(256), 0, st, Xh, N, S, Sp, F);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_spec_zero_pad(float* Xh, int N, int S, int F, hipStream_t st) { return launch_spec_zero_rows(Xh, N, S, spec_rows(S), F, st); }

int launch_nng(const float* A, int F, int Sp, int G, const float* Wq, size_t wstride, int nct, float* C, int num_cus, hipStream_t st,
               const char* tag, const float* bias, const float* gscale, size_t a_gstride) {
    const unsigned ags = (unsigned)(a_gstride != 0 ? a_gstride : (size_t)Sp * F);
    const size_t lds = nng_lds_bytes(nct / 4);
    const int RT = (Sp / 16) * G;
    int Gw = 2 * (num_cus > 0 ? num_cus : 256);
    if (Gw > ceil_div(RT, 8)) Gw = ceil_div(RT, 8);
    if (Gw < 1) Gw = 1;
    if (nct == 12) {
        EEG_SET_MAX_LDS((gemm_nng_kernel<3, 2>), lds);
        EEG_LAUNCH_P(tag, (gemm_nng_kernel<3, 2>), dim3(Gw), dim3(256), lds, st, A, ags, F, Sp, G, Wq, (unsigned)wstride, C, 16 * nct, bias, gscale);
    } else if (nct == 4) {
        EEG_SET_MAX_LDS((gemm_nng_kernel<1, 2>), lds);
        EEG_LAUNCH_P(tag, (gemm_nng_kernel<1, 2>), dim3(Gw), dim3(256), lds, st, A, ags, F, Sp, G, Wq, (unsigned)wstride, C, 16 * nct, bias, gscale);
    } else {
        return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

TngPlan tng_plan(int F, int Sp, int G, int num_cus) {
    TngPlan p{};
    if (F < 4 || F % 4 != 0 || Sp < 16 || Sp % 16 != 0 || G < 1) return p;
    if (F == 64) {
        p.planar = 1; p.KT = 2; p.nkb = 1;
    } else {                                               // per-lane source pointers, k-blocks of 4 or 5 tiles per wave slice: least padded K
        int best = 5, bcost = 1 << 30, bnkb = 1;
        for (int kt = 5; kt >= 4; --kt) {
            const int nkb = ceil_div(F, 32 * kt), cost = nkb * 32 * kt;
            if (cost < bcost) { best = kt; bcost = cost; bnkb = nkb; }
        }
        p.KT = best; p.nkb = bnkb;
    }
    const int target = 2 * (num_cus > 0 ? num_cus : 256);
    int spg = target / (G * p.nkb);
    if (spg < 1) spg = 1;
    int rps = round_up(ceil_div(Sp, spg), 16);
    if (rps < 64) rps = 64;
    if (rps > Sp) rps = Sp;
    p.rps = rps;
    p.spg = ceil_div(Sp, rps);
    p.ok = 1;
    return p;
}

namespace {
template <int KT, bool PLANAR>
int launch_tng_one(const TngPlan& p, const float* A, int F, int Sp, int G, const float* dY, float* partial, hipStream_t st, const char* tag,
                   long long skew) {
    constexpr int RC = 16, OT = 6;
    const size_t lds = 3 * (size_t)(RC * 32 * (KT + OT)) * sizeof(float);
    SegPtrs segs;
    for (int m = 0; m < kMaxM; ++m) segs.p[m] = m == 0 ? A : nullptr;
    EEG_SET_MAX_LDS((gemm_tnq_grouped_kernel<KT, OT, RC, PLANAR>), lds);
    EEG_LAUNCH_P(tag, (gemm_tnq_grouped_kernel<KT, OT, RC, PLANAR>), dim3(p.nkb, G * p.spg), dim3(256), lds, st, segs, F, Sp, G, p.spg, dY, 192,
                 0, 192, partial, p.rps, skew);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
}  // namespace

int launch_tng_pair(const TngPlan& p, const float* Ah, const float* Arh, int Sp, int G, const float* dY, float* part_g, float* part_c,
                    hipStream_t st, const char* tag, size_t ah_gstride) {
    if (!p.ok || !p.planar || p.KT != 2 || p.nkb != 1) return 1;
    const long long skew = ah_gstride != 0 ? (long long)ah_gstride - (long long)Sp * 64 : 0;
    constexpr int RC = 16, KT = 2;
    const size_t lds = 3 * (size_t)(RC * 32 * (KT + 4)) * sizeof(float);
    TnqJob ja, jb;
    for (int m = 0; m < kMaxM; ++m) { ja.segs.p[m] = m == 0 ? Ah : nullptr; jb.segs.p[m] = m == 0 ? Arh : nullptr; }
    ja.ycol0 = 0; ja.Ov = 128; ja.partial = part_g;
    jb.ycol0 = 128; jb.Ov = 64; jb.partial = part_c;
    EEG_SET_MAX_LDS((gemm_tnq_grouped_pair_kernel<KT, RC, true>), lds);
    EEG_LAUNCH_P(tag, (gemm_tnq_grouped_pair_kernel<KT, RC, true>), dim3(1, 2 * G * p.spg), dim3(256), lds, st, ja, jb, 64, Sp, G, p.spg, dY, 192,
                 p.rps, skew, (long long)0);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

namespace {
template <int KQ, bool SWZ>
int launch_nnf_one(const float* A, size_t ags, int K, int Sp, int G, const float* Wr, size_t wstride, float* C, int num_cus, hipStream_t st,
                   const char* tag, const float* bias, const float* gscale) {
    const size_t lds = nnf_lds_bytes(K);
    const int total = G * ceil_div(Sp, 64);
    int nb = 2 * (num_cus > 0 ? num_cus : 256);
    if (nb > total) nb = total;
    EEG_SET_MAX_LDS((gemm_nnf_kernel<KQ, SWZ>), lds);
    EEG_LAUNCH_P(tag, (gemm_nnf_kernel<KQ, SWZ>), dim3(nb), dim3(256), lds, st, A, (long long)ags, K, Sp, G, Wr, (unsigned)wstride, C, bias, gscale);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
}  // namespace
int launch_nnf(const float* A, size_t a_gstride, int K, int Sp, int G, const float* Wr, size_t wstride, float* C, int num_cus, hipStream_t st,
               const char* tag, const float* bias, const float* gscale) {
    if (K < 4 || K % 4 != 0 || Sp < 16 || G < 1 || (double)Sp * 192 * 4 >= 2.0e9) return -1;
    const size_t ags = a_gstride != 0 ? a_gstride : (size_t)Sp * K;
    const int kq = ceil_div(K, 8);
    const bool odd = ((K / 4) & 1) != 0;         // an odd number of 16-byte units per row: the plain row-major image is conflict-free
#define EEG_NNF(KQ, SWZ) return launch_nnf_one<KQ, SWZ>(A, ags, K, Sp, G, Wr, wstride, C, num_cus, st, tag, bias, gscale)
    if (K == 64) EEG_NNF(8, true);
    if (odd && kq == 13) EEG_NNF(13, false);
    if (odd && kq == 9) EEG_NNF(9, false);
    if (odd && kq == 5) EEG_NNF(5, false);
    if (odd && kq == 2) EEG_NNF(2, false);
#undef EEG_NNF
    return -1;
}

int launch_dxf(const float* dYh, int Sp, int S, int N, int Fin, const float* Wtq, size_t wstride, const float* basis, float* dX,
               hipStream_t st, const char* tag) {
    if (Fin != 64 || N < 1 || N > kDxfMaxN || Sp < 16 || S < 1 || S > Sp) return -1;
    if ((double)Sp * 192 * 4 >= 2147483648.0 || (double)S * N * 64 * 4 >= 2147483648.0) return -1;
    const size_t lds = dxf_lds_bytes();
    if (N == 19) {
        EEG_SET_MAX_LDS(gemm_dxf_kernel<19>, lds);
        EEG_LAUNCH_P(tag, gemm_dxf_kernel<19>, dim3(ceil_div(Sp, kDxfRows)), dim3(256), lds, st, dYh, Sp, S, N, Wtq, (unsigned)wstride, basis, dX);
    } else {
        EEG_SET_MAX_LDS(gemm_dxf_kernel<0>, lds);
        EEG_LAUNCH_P(tag, gemm_dxf_kernel<0>, dim3(ceil_div(Sp, kDxfRows)), dim3(256), lds, st, dYh, Sp, S, N, Wtq, (unsigned)wstride, basis, dX);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// Row splits of the fused kernel: every workgroup of the launch resident at once (2 per CU), rows per split a multiple of 16.
TnfPlan tnf_plan(int Fin, int H, int Sp, int G, int num_cus) {
    TnfPlan p{};
    if (H != 64 || Fin < 4 || Fin % 4 != 0 || Fin > 128 || Sp < 16 || Sp % 16 != 0 || G < 1) return p;
    if ((double)Sp * 192 * 4 >= 2147483648.0) return p;      // a frequency's rows go through 2-GB buffer descriptors (platform.h make_wbuf)
    p.fxt = ceil_div(Fin, 32);
#ifdef EEG_X_TNF_TARGET1
    const int target = 1 * (num_cus > 0 ? num_cus : 256);
#else
    const int target = tnf_wgs_per_cu(p.fxt) * (num_cus > 0 ? num_cus : 256);
#endif
    int spg = target / G;
    if (spg < 1) spg = 1;
    int rps = round_up(ceil_div(Sp, spg), 16);
    if (rps < 64) rps = 64;
    if (rps > Sp) rps = Sp;
    p.rps = rps;
    p.spg = ceil_div(Sp, rps);
    p.ok = 1;
    return p;
}
namespace {
template <int FXT>
int launch_tnf_one(const TnfPlan& p, const float* Xh, size_t xgs, int Fin, const float* Hh, size_t hgs, const float* RHh, const float* dY,
                   int Sp, int G, float* part_x, float* part_g, float* part_c, hipStream_t st, const char* tag) {
    const size_t lds = tnf_lds_bytes(FXT);
    EEG_SET_MAX_LDS((gemm_tnf_kernel<FXT>), lds);
    EEG_LAUNCH_P(tag, (gemm_tnf_kernel<FXT>), dim3(G * p.spg), dim3(256), lds, st, Xh, (long long)xgs, Fin, Hh, (long long)hgs, RHh, dY, Sp,
                 p.spg, p.rps, part_x, part_g, part_c);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
}  // namespace
int launch_tnf(const TnfPlan& p, const float* Xh, size_t x_gstride, int Fin, const float* Hh, size_t h_gstride, const float* RHh,
               const float* dY, int Sp, int G, float* part_x, float* part_g, float* part_c, hipStream_t st, const char* tag) {
    if (!p.ok) return 1;
    const size_t xgs = x_gstride != 0 ? x_gstride : (size_t)Sp * Fin, hgs = h_gstride != 0 ? h_gstride : (size_t)Sp * 64;
    switch (p.fxt) {
        case 1: return launch_tnf_one<1>(p, Xh, xgs, Fin, Hh, hgs, RHh, dY, Sp, G, part_x, part_g, part_c, st, tag);
        case 2: return launch_tnf_one<2>(p, Xh, xgs, Fin, Hh, hgs, RHh, dY, Sp, G, part_x, part_g, part_c, st, tag);
        case 3: return launch_tnf_one<3>(p, Xh, xgs, Fin, Hh, hgs, RHh, dY, Sp, G, part_x, part_g, part_c, st, tag);
        case 4: return launch_tnf_one<4>(p, Xh, xgs, Fin, Hh, hgs, RHh, dY, Sp, G, part_x, part_g, part_c, st, tag);
    }
    return 1;
}

int launch_tng(const TngPlan& p, const float* A, int F, int Sp, int G, const float* dY, float* partial, hipStream_t st, const char* tag,
               size_t a_gstride) {
    if (!p.ok) return 1;
    const long long skew = a_gstride != 0 ? (long long)a_gstride - (long long)Sp * F : 0;
    if (p.planar) return launch_tng_one<2, true>(p, A, F, Sp, G, dY, partial, st, tag, skew);
    if (p.KT == 4) return launch_tng_one<4, false>(p, A, F, Sp, G, dY, partial, st, tag, skew);
    return launch_tng_one<5, false>(p, A, F, Sp, G, dY, partial, st, tag, skew);
}

}  // namespace eeg
