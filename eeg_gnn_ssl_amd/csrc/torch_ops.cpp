// TORCH_LIBRARY(eeg_dcrnn_cpp): the operator-level boundary of SURVEY.md §8(b) for C++ callers and for eager launches
// that should not pass through Python / ctypes.  A thin shim over the C ABI of libeeg_dcrnn_hip.so (include/eeg_dcrnn.h):
// argument checks, output / scratch allocation from torch's caching allocator, the current HIP stream.  No compute, no
// CPU kernels: the schemas are registered for the CUDA (= HIP) dispatch key only.  Built separately from the kernels
// (g++ against the torch headers, no device code): __graft_entry__.build() -> eeg_gnn_ssl_amd/libeeg_dcrnn_torch.so;
// load with torch.ops.load_library() (Python: eeg_gnn_ssl_amd.native_ops.load()) or link it into a C++ program.
// The Python operator library `torch.ops.eeg_dcrnn.*` (ops.py) remains the one with the autograd formulas.
//
// Reference code each operator replaces: hop_polys -- cell.py:83-93 (the Chebyshev-style recursion on the supports);
// diffusion_hops -- the same recursion applied to the features (north_star's HBM-bound step); dconv / dconv_bwd --
// DiffusionGraphConv.forward, cell.py:66-118, and autograd's replay of it; pack_cell -- the parameters of one DCGRUCell
// (cell.py:160-175) in the fragment order the recurrent kernels read.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/eeg_dcrnn.h"

namespace {

void* cur_stream() { return static_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }

const float* fp(const at::Tensor& t) { return t.data_ptr<float>(); }

at::Tensor checked(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), "eeg_dcrnn_cpp: ", name, " must be a GPU tensor (there is no CPU path)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, "eeg_dcrnn_cpp: ", name, " must be float32");
    return t.contiguous();
}
void ok(int rc, const char* what) { TORCH_CHECK(rc == 0, what, ": ", eeg_dcrnn_last_error()); }
at::Tensor scratch(size_t floats, const at::Tensor& like) {
    return at::empty({static_cast<int64_t>(floats)}, like.options());
}

// supports: S tensors (N,N) or (B,N,N) -> P (G, S*K, N, N), G = batch if any support is per-clip else 1
at::Tensor hop_polys(at::TensorList supports, int64_t K, int64_t batch) {
    TORCH_CHECK(!supports.empty(), "eeg_dcrnn_cpp::hop_polys: no supports");
    const int64_t n = supports[0].size(-1);
    bool batched = false;
    for (const auto& s : supports) batched = batched || s.dim() == 3;
    std::vector<at::Tensor> keep;
    std::vector<const float*> ptrs;
    for (const auto& s0 : supports) {
        TORCH_CHECK(s0.size(-1) == n && s0.size(-2) == n, "eeg_dcrnn_cpp::hop_polys: supports must be (..., N, N)");
        at::Tensor s = s0;
        if (batched && s.dim() == 2) s = s.unsqueeze(0).expand({batch, n, n});
        TORCH_CHECK(s.dim() == 2 || s.size(0) == batch, "eeg_dcrnn_cpp::hop_polys: support batch != input batch");
        keep.push_back(checked(s, "supports"));
        ptrs.push_back(fp(keep.back()));
    }
    const int64_t g = batched ? batch : 1;
    at::Tensor out = at::empty({g, static_cast<int64_t>(keep.size()) * K, n, n}, keep[0].options());
    ok(eeg_dcrnn_hop_polys(ptrs.data(), static_cast<int>(keep.size()), static_cast<int>(g), static_cast<int>(n),
                           static_cast<int>(K), out.data_ptr<float>(), cur_stream()), "eeg_dcrnn_cpp::hop_polys");
    return out;
}

// x (S,N,F), P (G,M-1,N,N) -> (M-1,S,N,F)
at::Tensor diffusion_hops(const at::Tensor& x_, const at::Tensor& p_, int64_t p_batched, int64_t batch) {
    const at::Tensor x = checked(x_, "x"), p = checked(p_, "P");
    TORCH_CHECK(x.dim() == 3 && p.dim() == 4, "eeg_dcrnn_cpp::diffusion_hops: x (S,N,F), P (G,M-1,N,N)");
    const int64_t s = x.size(0), n = x.size(1), f = x.size(2), m = p.size(1) + 1;
    at::Tensor out = at::empty({m - 1, s, n, f}, x.options());
    ok(eeg_dcrnn_diffuse_fwd(fp(x), fp(p), static_cast<int>(p_batched), static_cast<int>(s), static_cast<int>(batch),
                             static_cast<int>(n), static_cast<int>(f), static_cast<int>(m), out.data_ptr<float>(), cur_stream()),
       "eeg_dcrnn_cpp::diffusion_hops");
    return out;
}

// x (B,N,F), P (G,M-1,N,N), weight ((F*M),O), biases (O) -> (B,N,O)
at::Tensor dconv(const at::Tensor& x_, const at::Tensor& p_, int64_t p_batched, const at::Tensor& w_, const at::Tensor& b_) {
    const at::Tensor x = checked(x_, "inputs_and_state"), p = checked(p_, "P"), w = checked(w_, "weight"), b = checked(b_, "biases");
    const int64_t bn = x.size(0), n = x.size(1), f = x.size(2), m = p.size(1) + 1, o = w.size(1);
    TORCH_CHECK(w.size(0) == f * m, "eeg_dcrnn_cpp::dconv: weight has ", w.size(0), " rows, expected (input_dim+hid_dim)*num_matrices = ", f * m);
    at::Tensor out = at::empty({bn, n, o}, x.options());
    at::Tensor ws = scratch(eeg_dcrnn_dconv_fwd_ws_floats(static_cast<int>(bn), static_cast<int>(n), static_cast<int>(f),
                                                          static_cast<int>(m), static_cast<int>(o)), x);
    ok(eeg_dcrnn_dconv_fwd(fp(x), fp(p), static_cast<int>(p_batched), static_cast<int>(bn), static_cast<int>(n), static_cast<int>(f),
                           static_cast<int>(m), fp(w), fp(b), static_cast<int>(o), out.data_ptr<float>(), ws.data_ptr<float>(),
                           cur_stream()), "eeg_dcrnn_cpp::dconv");
    return out;
}

// -> (dx (B,N,F) or empty, dweight ((F*M),O), dbiases (O))
std::tuple<at::Tensor, at::Tensor, at::Tensor> dconv_bwd(const at::Tensor& dout_, const at::Tensor& x_, const at::Tensor& p_,
                                                         int64_t p_batched, const at::Tensor& w_, bool need_dx) {
    const at::Tensor dout = checked(dout_, "grad_output"), x = checked(x_, "inputs_and_state"), p = checked(p_, "P"),
                     w = checked(w_, "weight");
    const int64_t bn = x.size(0), n = x.size(1), f = x.size(2), m = p.size(1) + 1, o = w.size(1);
    at::Tensor dx = need_dx ? at::empty_like(x) : at::empty({0}, x.options());
    at::Tensor dw = at::empty({f * m, o}, x.options()), db = at::empty({o}, x.options());
    at::Tensor ws = scratch(eeg_dcrnn_dconv_bwd_ws_floats(static_cast<int>(bn), static_cast<int>(n), static_cast<int>(f),
                                                          static_cast<int>(m), static_cast<int>(o)), x);
    ok(eeg_dcrnn_dconv_bwd(fp(x), fp(p), static_cast<int>(p_batched), static_cast<int>(bn), static_cast<int>(n), static_cast<int>(f),
                           static_cast<int>(m), fp(w), static_cast<int>(o), fp(dout), need_dx ? dx.data_ptr<float>() : nullptr,
                           dw.data_ptr<float>(), db.data_ptr<float>(), ws.data_ptr<float>(), cur_stream()),
       "eeg_dcrnn_cpp::dconv_bwd");
    return std::make_tuple(dx, dw, db);
}

at::Tensor pack_cell(const at::Tensor& wg_, const at::Tensor& bg_, const at::Tensor& wc_, const at::Tensor& bc_, int64_t fin,
                     int64_t h, int64_t m) {
    const at::Tensor wg = checked(wg_, "dconv_gate.weight"), bg = checked(bg_, "dconv_gate.biases"),
                     wc = checked(wc_, "dconv_candidate.weight"), bc = checked(bc_, "dconv_candidate.biases");
    const int64_t rows = (fin + h) * m;
    TORCH_CHECK(wg.size(0) == rows && wg.size(1) == 2 * h && wc.size(0) == rows && wc.size(1) == h && bg.numel() == 2 * h &&
                    bc.numel() == h, "eeg_dcrnn_cpp::pack_cell: parameter shapes do not match input_dim / num_units / num_matrices");
    at::Tensor pack = scratch(eeg_dcrnn_pack_floats(static_cast<int>(fin), static_cast<int>(h), static_cast<int>(m)), wg);
    ok(eeg_dcrnn_pack_cell(fp(wg), fp(bg), fp(wc), fp(bc), static_cast<int>(fin), static_cast<int>(h), static_cast<int>(m),
                           pack.data_ptr<float>(), cur_stream()), "eeg_dcrnn_cpp::pack_cell");
    return pack;
}

}  // namespace

TORCH_LIBRARY(eeg_dcrnn_cpp, m) {
    m.def("hop_polys(Tensor[] supports, int max_diffusion_step, int batch) -> Tensor");
    m.def("diffusion_hops(Tensor x, Tensor P, int p_batched, int batch) -> Tensor");
    m.def("dconv(Tensor x, Tensor P, int p_batched, Tensor weight, Tensor biases) -> Tensor");
    m.def("dconv_bwd(Tensor dout, Tensor x, Tensor P, int p_batched, Tensor weight, bool need_dx) -> (Tensor, Tensor, Tensor)");
    m.def("pack_cell(Tensor wg, Tensor bg, Tensor wc, Tensor bc, int fin, int h, int m) -> Tensor");
}
TORCH_LIBRARY_IMPL(eeg_dcrnn_cpp, CUDA, m) {
    m.impl("hop_polys", &hop_polys);
    m.impl("diffusion_hops", &diffusion_hops);
    m.impl("dconv", &dconv);
    m.impl("dconv_bwd", &dconv_bwd);
    m.impl("pack_cell", &pack_cell);
}
