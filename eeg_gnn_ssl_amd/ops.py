"""The `torch.ops.eeg_dcrnn.*` operator library over the C ABI (include/eeg_dcrnn.h).

Every operator of the DCRNN hot path is registered with the PyTorch dispatcher (`torch.library`):
schema, device implementation (tensors in -> C-ABI launch on torch's current stream -> tensors out),
a fake (meta) implementation for tracing, and — for the differentiable ones — the autograd formula,
which calls the matching `*_bwd` operator.  PyTorch is plumbing here (device memory, streams, autograd
bookkeeping); all arithmetic runs in the HIP kernels of libeeg_dcrnn_hip.so.  The implementations are
registered for the CUDA (= HIP on ROCm) key only: this package has no CPU path (a CPU tensor gets the
dispatcher's "no kernel for the CPU backend" error, and the C-ABI stub refuses non-GPU pointers).

Operators (namespace `eeg_dcrnn`):
    hop_polys, pack_cell, diffusion_hops, dconv (+ dconv_bwd), dcgru_layer (+ dcgru_layer_bwd),
    dcgru_decoder (+ dcgru_decoder_bwd), cls_head (+ cls_head_bwd), rng_take_, dropout_mask, gather_last, corr_graph,
    fft_features, bce_logits, ce_logits, masked_loss, cls_head_loss, pack_cells, clip_adam_, clip_adam_dev_, teacher_flags_,
    augment_draw_.
The functions below them are the Python conveniences the modules in model/ and train_step.py call.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import DecoderDims, LayerDims

ACT_CODES = {"tanh": 0, "relu": 1, None: 1}   # the reference maps anything but 'tanh' to relu (cell.py:146)
NS = "eeg_dcrnn"
_libdef = torch.library.Library(NS, "DEF")


# OPT-IN arithmetic of the two hoisted NN GEMMs of a DCGRU layer (x-part pre-activations, input gradient): 0 = the fp32 matrix pipe
# (the default and the contract's arithmetic), 1 = three-term bf16 split (6 of 9 partial products on v_mfma_f32_16x16x32_bf16,
# fp32 accumulation; include/eeg_dcrnn.h `eeg_layer_dims.pack3`).  Set it BEFORE building / capturing a step (`set_gemm_mode`, or
# EEG_DCRNN_SPLIT_BF16=1 in the environment at import): a cell's weight pack then carries the bf16 term packs behind its fp32 part,
# and forward and backward of a step must run under the same mode.
import os as _os

GEMM_MODE = 1 if _os.environ.get("EEG_DCRNN_SPLIT_BF16", "0") not in ("", "0") else 0


def set_gemm_mode(mode: int) -> int:
    """0 = fp32 MFMA (default), 1 = three-term bf16 split of the hoisted NN GEMMs (64 units; other sizes keep fp32). Returns the
    previous mode."""
    global GEMM_MODE
    if mode not in (0, 1):
        raise ValueError("gemm mode must be 0 (fp32 MFMA) or 1 (three-term bf16 split)")
    prev, GEMM_MODE = GEMM_MODE, int(mode)
    return prev


def _pack3_halves(fin, h, m) -> int:
    return int(_lib.get_lib().query("eeg_dcrnn_pack3_halves", int(fin), int(h), int(m))) if GEMM_MODE else 0


def _pack_base_floats(fin, h, m) -> int:
    """floats of the fp32 part of a cell's pack, rounded up so that what follows is 16-byte aligned"""
    return (int(_lib.get_lib().query("eeg_dcrnn_pack_floats", int(fin), int(h), int(m))) + 3) // 4 * 4


def _set_pack3(dims, pack, fin, h, m):
    """point dims.pack3 at the bf16 term packs behind the fp32 part of `pack` -- decided by what THIS pack tensor carries (it was
    built under the mode of its forward call), not by the mode at the time of the call: a backward that runs after `set_gemm_mode`
    changed keeps the arithmetic of its forward and never addresses past the end of a pack built without the term packs."""
    base = _pack_base_floats(fin, h, m)
    halves = int(_lib.get_lib().query("eeg_dcrnn_pack3_halves", int(fin), int(h), int(m)))
    if halves > 0 and pack.numel() >= base + (halves + 1) // 2:
        dims.pack3 = pack.data_ptr() + 4 * base
    elif pack.numel() != base:
        raise RuntimeError(f"weight pack has {pack.numel()} floats, expected {base} (fp32) or {base + (halves + 1) // 2} (with the "
                           f"bf16 term packs) for input_dim={fin}, num_units={h}, num_matrices={m}")


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(t: torch.Tensor):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return ctypes.c_void_p(0)


def _check(lib, t: torch.Tensor, name: str, dtype=torch.float32):
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if lib.is_device_build and not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; eeg_gnn_ssl_amd runs only on an MI355X (HIP) device "
                           f"and has no CPU path")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: tensor must be contiguous")


def _check_polys(p: torch.Tensor, p_batched: int, b: int, n: int, m: int, what: str = "P"):
    """hop-polynomial tensor of `ops.hop_polys`: (G, M-1, N, N) with G = B (per-clip graphs) or 1 (shared) -- the kernels index it
    with exactly these extents, so a tensor built for another node count / batch / hop count is refused here, not read out of bounds"""
    want = (b if p_batched else 1, m - 1, n, n)
    if m > 1 and tuple(p.shape) != want:
        raise RuntimeError(f"{what} has shape {tuple(p.shape)}, expected {want} (graphs, hop matrices - 1, num_nodes, num_nodes)")


def _check_lengths(lengths: torch.Tensor, b: int):
    if lengths.numel() != b:
        raise RuntimeError(f"seq_lengths has {lengths.numel()} entries for a batch of {b}")


def _new(shape, like: torch.Tensor, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _none_if_empty(t: Optional[torch.Tensor]):
    return None if (t is None or t.numel() == 0) else t


_impls = {}       # operator name -> device implementation (read by tests/emu_support.py, which drives the same
#                   implementations through the emulator build of the kernel sources)


def _tensor_device(args):
    for a in args:
        if isinstance(a, torch.Tensor):
            if a.is_cuda:
                return a.device
        elif isinstance(a, (list, tuple)):
            d = _tensor_device(a)
            if d is not None:
                return d
    return None


def _device_guarded(impl):
    """Python-registered operators get no DeviceGuard from the dispatcher: a call on tensors of cuda:1 while cuda:0 is the current
    device would allocate and launch on the wrong GPU.  (One process per GPU -- the layout this package is built for -- never takes
    the slow branch: one integer comparison per call.)"""
    def run(*args, **kwargs):
        dev = _tensor_device(args) or _tensor_device(tuple(kwargs.values()))
        if dev is None or dev.index is None or dev.index == torch.cuda.current_device():
            return impl(*args, **kwargs)
        with torch.cuda.device(dev):
            return impl(*args, **kwargs)
    run.__name__ = getattr(impl, "__name__", "impl")
    return run


def _define(name: str, schema: str, impl, fake):
    """schema + device implementation (CUDA key = HIP) + fake implementation."""
    _libdef.define(f"{name}{schema}")
    _libdef.impl(name, _device_guarded(impl), "CUDA")
    _impls[name] = impl
    torch.library.register_fake(f"{NS}::{name}", fake, lib=_libdef)


def _zero_impl(t) -> None:
    """t <- 0 as ONE memset on the stream (no framework fill kernel in the captured step)"""
    lib = _lib.get_lib()
    if not t.is_contiguous():
        raise RuntimeError("zero_: tensor must be contiguous")
    if lib.is_device_build and not t.is_cuda:
        raise RuntimeError("zero_: eeg_gnn_ssl_amd runs only on an MI355X (HIP) device and has no CPU path")
    lib.call("eeg_dcrnn_zero", _p(t), t.numel() * t.element_size(), _stream(t))


_define("zero_", "(Tensor(a!) t) -> ()", _zero_impl, lambda t: None)


def zero_(t: torch.Tensor) -> torch.Tensor:
    torch.ops.eeg_dcrnn.zero_(t)
    return t


def num_matrices(filter_type: str, max_diffusion_step: int) -> int:
    """cell.py:35,151-158."""
    return (2 if filter_type == "dual_random_walk" else 1) * max_diffusion_step + 1


# =============================================================================================
# hop polynomials, weight packs, the diffusion step
# =============================================================================================
def _hop_polys_impl(supports: List[torch.Tensor], max_diffusion_step: int, batch: int) -> torch.Tensor:
    lib = _lib.get_lib()
    sups = list(supports)
    if len(sups) == 0:
        raise RuntimeError("hop_polys: empty supports list")
    batched = any(s.dim() == 3 for s in sups)
    n = sups[0].shape[-1]
    norm = []
    for i, s in enumerate(sups):
        if s.shape[-1] != n or s.shape[-2] != n:
            raise RuntimeError(f"supports[{i}] has shape {tuple(s.shape)}, expected (..., {n}, {n})")
        if s.dim() == 3 and s.shape[0] != batch:
            raise RuntimeError(f"supports[{i}] batch {s.shape[0]} != input batch {batch}")
        if batched and s.dim() == 2:
            s = s.unsqueeze(0).expand(batch, n, n)
        s = s.to(torch.float32).contiguous()
        _check(lib, s, f"supports[{i}]")
        norm.append(s)
    g = batch if batched else 1
    out = _new((g, len(norm) * max_diffusion_step, n, n), norm[0])
    arr = (ctypes.c_void_p * len(norm))(*[s.data_ptr() for s in norm])
    lib.call("eeg_dcrnn_hop_polys", arr, len(norm), g, n, max_diffusion_step, _p(out), _stream(out))
    return out


def _hop_polys_fake(supports, max_diffusion_step, batch):
    g = batch if any(s.dim() == 3 for s in supports) else 1
    n = supports[0].shape[-1]
    return supports[0].new_empty((g, len(supports) * max_diffusion_step, n, n), dtype=torch.float32)


_define("hop_polys", "(Tensor[] supports, int max_diffusion_step, int batch) -> Tensor", _hop_polys_impl, _hop_polys_fake)


def _pack_cell_impl(wg, bg, wc, bc, fin: int, h: int, m: int) -> torch.Tensor:
    lib = _lib.get_lib()
    tensors = [wg.detach(), bg.detach(), wc.detach(), bc.detach()]
    for t, nm in zip(tensors, ("dconv_gate.weight", "dconv_gate.biases", "dconv_candidate.weight", "dconv_candidate.biases")):
        _check(lib, t, nm)
    rows = (fin + h) * m
    if tuple(wg.shape) != (rows, 2 * h) or tuple(wc.shape) != (rows, h) or tuple(bg.shape) != (2 * h,) or tuple(bc.shape) != (h,):
        raise RuntimeError(f"cell parameter shapes {tuple(wg.shape)}, {tuple(bg.shape)}, {tuple(wc.shape)}, {tuple(bc.shape)} "
                           f"do not match input_dim={fin}, num_units={h}, num_matrices={m}")
    base = _pack_base_floats(fin, h, m)
    halves = _pack3_halves(fin, h, m)
    pack = _new((base + (halves + 1) // 2,), wg)
    lib.call("eeg_dcrnn_pack_cell", _p(tensors[0]), _p(tensors[1]), _p(tensors[2]), _p(tensors[3]), fin, h, m, _p(pack), _stream(pack))
    if halves > 0:      # opt-in bf16 split: the three-term packs of the x-part weights ride behind the fp32 packs
        lib.call("eeg_dcrnn_pack_cell_bf16x3", _p(tensors[0]), _p(tensors[2]), fin, h, m, ctypes.c_void_p(pack.data_ptr() + 4 * base), _stream(pack))
    return pack


def _pack_floats(fin, h, m):
    """size of a cell's weight pack for shape inference: `eeg_dcrnn_pack_floats` is a pure host computation of the library
    (make_cell_pack, csrc/kernels_pack.h), so the fake implementations ask it instead of mirroring its layout"""
    return _pack_base_floats(fin, h, m) + (_pack3_halves(fin, h, m) + 1) // 2


_define("pack_cell", "(Tensor wg, Tensor bg, Tensor wc, Tensor bc, int fin, int h, int m) -> Tensor", _pack_cell_impl,
        lambda wg, bg, wc, bc, fin, h, m: wg.new_empty((_pack_floats(fin, h, m),)))


# ---- spectral form of the hoisted x-part for ONE shared symmetric support (include/eeg_dcrnn.h: eeg_layer_dims.spectral) ----------
# 1 (default): layers whose supports are given as ONE 2-D (N,N) tensor -- i.e. declared shared by the batch -- and turn out
# symmetric run their hoisted x-part in the eigenbasis of the support (K = Fin instead of M*Fin); 0: always the general path.
SPECTRAL_MODE = 0 if _os.environ.get("EEG_DCRNN_SPECTRAL", "1") in ("", "0") else 1
SPECTRAL_TOL = 2e-6          # largest |S - U diag(lam) U^T| (relative to max |S|) for which a support counts as symmetric


def set_spectral_mode(mode: int) -> int:
    """0 = general path everywhere, 1 = spectral form for shared symmetric supports (default).  Returns the previous mode."""
    global SPECTRAL_MODE
    prev, SPECTRAL_MODE = SPECTRAL_MODE, 1 if mode else 0
    return prev


def _spectral_basis_impl(support) -> torch.Tensor:
    lib = _lib.get_lib()
    sup = support.to(torch.float32).contiguous()
    _check(lib, sup, "support")
    if sup.dim() != 2 or sup.shape[0] != sup.shape[1]:
        raise RuntimeError(f"spectral_basis: support has shape {tuple(sup.shape)}, expected (num_nodes, num_nodes)")
    n = sup.shape[0]
    out = _new((lib.query("eeg_dcrnn_spectral_basis_floats", n),), sup)
    lib.call("eeg_dcrnn_spectral_basis", _p(sup), n, _p(out), _stream(sup))
    return out


def _spectral_basis_floats(n):
    return int(_lib.get_lib().query("eeg_dcrnn_spectral_basis_floats", int(n)))


_define("spectral_basis", "(Tensor support) -> Tensor", _spectral_basis_impl,
        lambda support: support.new_empty((_spectral_basis_floats(support.shape[0]),), dtype=torch.float32))


def _pack_cell_spectral_impl(wg, wc, basis, fin: int, h: int, m: int, n: int) -> torch.Tensor:
    lib = _lib.get_lib()
    wg, wc = wg.detach(), wc.detach()
    for t, nm in ((wg, "dconv_gate.weight"), (wc, "dconv_candidate.weight"), (basis, "basis")):
        _check(lib, t, nm)
    rows = (fin + h) * m
    if tuple(wg.shape) != (rows, 2 * h) or tuple(wc.shape) != (rows, h) or basis.numel() != _spectral_basis_floats(n):
        raise RuntimeError(f"pack_cell_spectral: operands {tuple(wg.shape)}, {tuple(wc.shape)}, basis {tuple(basis.shape)} do not match "
                           f"input_dim={fin}, num_units={h}, num_matrices={m}, num_nodes={n}")
    size = lib.query("eeg_dcrnn_spectral_pack_floats", fin, h, m, n)
    if size == 0:
        raise RuntimeError(f"pack_cell_spectral: the spectral form exists for 64 units (got num_units={h}, input_dim={fin})")
    out = _new((size,), wg)
    lib.call("eeg_dcrnn_pack_cell_spectral", _p(wg), _p(wc), _p(basis), fin, h, m, n, _p(out), _stream(out))
    return out


_define("pack_cell_spectral", "(Tensor wg, Tensor wc, Tensor basis, int fin, int h, int m, int n) -> Tensor", _pack_cell_spectral_impl,
        lambda wg, wc, basis, fin, h, m, n: wg.new_empty((int(_lib.get_lib().query("eeg_dcrnn_spectral_pack_floats", fin, h, m, n)),)))


def _pack_cells_impl(wgs, bgs, wcs, bcs, h: int, m: int, basis, n: int):
    """the packs of all cells of an encoder in ONE launch: [pack_0 .. pack_{L-1}] + (with a basis) [spack_0 .. spack_{L-1}]"""
    lib = _lib.get_lib()
    cells = len(wgs)
    if not (cells == len(bgs) == len(wcs) == len(bcs)) or cells < 1 or cells > 4:
        raise RuntimeError(f"pack_cells: {cells} cells (1..4, one weight / bias tensor of each kind per cell)")
    wgs, bgs, wcs, bcs = ([t.detach() for t in ts] for ts in (wgs, bgs, wcs, bcs))
    fins, packs, spacks = [], [], []
    for c in range(cells):
        fin = wgs[c].shape[0] // m - h
        rows = (fin + h) * m
        for t, nm in ((wgs[c], "dconv_gate.weight"), (bgs[c], "dconv_gate.biases"), (wcs[c], "dconv_candidate.weight"), (bcs[c], "dconv_candidate.biases")):
            _check(lib, t, nm)
        if fin < 4 or tuple(wgs[c].shape) != (rows, 2 * h) or tuple(wcs[c].shape) != (rows, h) or tuple(bgs[c].shape) != (2 * h,) \
                or tuple(bcs[c].shape) != (h,):
            raise RuntimeError(f"pack_cells: cell {c} parameter shapes {tuple(wgs[c].shape)}, {tuple(bgs[c].shape)}, {tuple(wcs[c].shape)}, "
                               f"{tuple(bcs[c].shape)} do not match num_units={h}, num_matrices={m}")
        if _pack3_halves(fin, h, m) > 0:
            raise RuntimeError("pack_cells: the opt-in bf16 split packs its cells one by one (ops.pack_cell)")
        fins.append(fin)
        packs.append(_new((_pack_base_floats(fin, h, m),), wgs[c]))
        if basis is not None:
            size = lib.query("eeg_dcrnn_spectral_pack_floats", fin, h, m, n)
            if size == 0:
                raise RuntimeError(f"pack_cells: the spectral form exists for 64 units (got num_units={h}, input_dim={fin})")
            spacks.append(_new((size,), wgs[c]))
    if basis is not None:
        _check(lib, basis, "basis")
        if basis.numel() != _spectral_basis_floats(n):
            raise RuntimeError(f"pack_cells: basis {tuple(basis.shape)} is not the block of a {n}-node support")
    tab = lambda ts: (ctypes.c_void_p * cells)(*[t.data_ptr() for t in ts])      # noqa: E731
    fin_arr = (ctypes.c_int32 * cells)(*fins)
    lib.call("eeg_dcrnn_pack_cells", cells, tab(wgs), tab(bgs), tab(wcs), tab(bcs), fin_arr, h, m, tab(packs), _p(basis), n,
             tab(spacks) if basis is not None else None, _stream(packs[0]))
    return packs + spacks


def _pack_cells_fake(wgs, bgs, wcs, bcs, h, m, basis, n):
    fins = [w.shape[0] // m - h for w in wgs]
    out = [wgs[0].new_empty((_pack_floats(f, h, m),)) for f in fins]
    if basis is not None:
        out += [wgs[0].new_empty((int(_lib.get_lib().query("eeg_dcrnn_spectral_pack_floats", f, h, m, n)),)) for f in fins]
    return out


_define("pack_cells", "(Tensor[] wg, Tensor[] bg, Tensor[] wc, Tensor[] bc, int h, int m, Tensor? basis, int n) -> Tensor[]",
        _pack_cells_impl, _pack_cells_fake)


def _diffusion_hops_impl(x, p, p_batched: int, batch: int) -> torch.Tensor:
    lib = _lib.get_lib()
    _check(lib, x, "x")
    _check(lib, p, "P")
    s, n, f = x.shape
    if p.dim() != 4:
        raise RuntimeError(f"P has shape {tuple(p.shape)}, expected (graphs, hop matrices - 1, num_nodes, num_nodes)")
    m = p.shape[1] + 1
    _check_polys(p, p_batched, batch, n, m)
    if batch < 1 or s % max(batch, 1) != 0:
        raise RuntimeError(f"diffusion_hops: {s} samples are not a multiple of the batch {batch}")
    out = _new((m - 1, s, n, f), x)
    lib.call("eeg_dcrnn_diffuse_fwd", _p(x), _p(p), p_batched, s, batch, n, f, m, _p(out), _stream(x))
    return out


_define("diffusion_hops", "(Tensor x, Tensor P, int p_batched, int batch) -> Tensor", _diffusion_hops_impl,
        lambda x, p, p_batched, batch: x.new_empty((p.shape[1],) + tuple(x.shape)))


# =============================================================================================
# DiffusionGraphConv (cell.py:66-118), differentiable
# =============================================================================================
def _dconv_impl(x, p, p_batched: int, weight, biases) -> torch.Tensor:
    lib = _lib.get_lib()
    x = x.contiguous()
    w, bvec = weight.detach().contiguous(), biases.detach().contiguous()
    for t, nm in ((x, "inputs_and_state"), (p, "P"), (w, "weight"), (bvec, "biases")):
        _check(lib, t, nm)
    b, n, f = x.shape
    if p.dim() != 4:
        raise RuntimeError(f"P has shape {tuple(p.shape)}, expected (graphs, hop matrices - 1, num_nodes, num_nodes)")
    m, o = p.shape[1] + 1, w.shape[1]
    _check_polys(p, p_batched, b, n, m)
    if w.shape[0] != f * m:
        raise RuntimeError(f"weight has {w.shape[0]} rows, expected (input_dim+hid_dim)*num_matrices = {f * m}")
    if bvec.numel() != o:
        raise RuntimeError(f"biases has {bvec.numel()} entries, expected output_dim = {o}")
    out = _new((b, n, o), x)
    ws = _new((lib.query("eeg_dcrnn_dconv_fwd_ws_floats", b, n, f, m, o),), x)
    lib.call("eeg_dcrnn_dconv_fwd", _p(x), _p(p), p_batched, b, n, f, m, _p(w), _p(bvec), o, _p(out), _p(ws), _stream(x))
    return out


def _dconv_bwd_impl(dout, x, p, p_batched: int, weight, need_dx: bool):
    lib = _lib.get_lib()
    dout, x, w = dout.contiguous(), x.contiguous(), weight.detach().contiguous()
    for t, nm in ((dout, "grad_output"), (x, "inputs_and_state"), (p, "P"), (w, "weight")):
        _check(lib, t, nm)
    b, n, f = x.shape
    m, o = p.shape[1] + 1, w.shape[1]
    _check_polys(p, p_batched, b, n, m)
    if w.shape[0] != f * m or tuple(dout.shape) != (b, n, o):
        raise RuntimeError(f"dconv_bwd: weight {tuple(w.shape)} / grad_output {tuple(dout.shape)} do not match inputs {tuple(x.shape)} with {m} hop matrices")
    dx = _new((b, n, f), x) if need_dx else _new((0,), x)
    dw, db = _new((f * m, o), x), _new((o,), x)
    ws = _new((lib.query("eeg_dcrnn_dconv_bwd_ws_floats", b, n, f, m, o),), x)
    lib.call("eeg_dcrnn_dconv_bwd", _p(x), _p(p), p_batched, b, n, f, m, _p(w), o, _p(dout), _p(dx) if need_dx else None,
             _p(dw), _p(db), _p(ws), _stream(x))
    return dx, dw, db


_define("dconv", "(Tensor x, Tensor P, int p_batched, Tensor weight, Tensor biases) -> Tensor", _dconv_impl,
        lambda x, p, p_batched, weight, biases: x.new_empty((x.shape[0], x.shape[1], weight.shape[1])))
_define("dconv_bwd", "(Tensor dout, Tensor x, Tensor P, int p_batched, Tensor weight, bool need_dx) -> (Tensor, Tensor, Tensor)",
        _dconv_bwd_impl,
        lambda dout, x, p, p_batched, weight, need_dx: (x.new_empty(x.shape if need_dx else (0,)), torch.empty_like(weight),
                                                        x.new_empty((weight.shape[1],))))


def _dconv_setup(ctx, inputs, output):
    x, p, p_batched, weight, _ = inputs
    ctx.save_for_backward(x, p, weight)
    ctx.p_batched = p_batched


def _dconv_backward(ctx, dout):
    x, p, weight = ctx.saved_tensors
    dx, dw, db = torch.ops.eeg_dcrnn.dconv_bwd(dout, x, p, ctx.p_batched, weight, ctx.needs_input_grad[0])
    return (dx if ctx.needs_input_grad[0] else None), None, None, dw, db


torch.library.register_autograd(f"{NS}::dconv", _dconv_backward, setup_context=_dconv_setup, lib=_libdef)


# =============================================================================================
# one DCGRU layer over a whole sequence (model.py:93-96 around cell.py:182-210)
# =============================================================================================
class GradSink:
    """Lets the backward operators write parameter gradients straight into caller-owned buffers (TrainStep's flat gradient
    bucket) instead of returning fresh tensors that autograd then adds into `p.grad` with one small kernel per parameter.

    No process-global state: the sink belongs to the object that owns the buffers (train_step.FlatParameters creates one and
    attaches it to each of ITS parameters as `param._eeg_grad_sink = (sink, index)`); a backward formula finds it through the
    parameter it was handed.  It is armed only inside `with sink: ...` (one backward pass); a parameter that receives a second
    gradient in the same pass falls back to the returned-tensor path (autograd accumulates), so results never depend on it."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self.params = list(params)
        self.armed = False
        self.written = [False] * len(self.params)
        for i, p in enumerate(self.params):
            p._eeg_grad_sink = (self, i)

    def __enter__(self):
        self.armed = True
        self.written = [False] * len(self.params)
        return self

    def __exit__(self, *exc):
        self.armed = False
        return False

    @staticmethod
    def take(param: torch.Tensor) -> Optional[torch.Tensor]:
        """the buffer to write `param`'s gradient into, or None (-> allocate and return it to autograd)"""
        ref = getattr(param, "_eeg_grad_sink", None)
        if ref is None:
            return None
        sink, i = ref
        tgt = param.grad
        if (not sink.armed or sink.written[i] or sink.params[i] is not param or tgt is None or not tgt.is_contiguous()
                or tgt.shape != param.shape):
            return None
        sink.written[i] = True
        return tgt


def _layer_dims(t_len, b, n, h, fin, m, act, p_batched, planes_ready):
    dims = LayerDims(t_len, b, n, h, fin, m, act, p_batched)
    if planes_ready:           # the layer below's Hplanes (M-1, T+1, B, N, Fin): slots 1..T are P_m x
        dims.x_planes_ready = 1
        dims.x_plane_stride = (t_len + 1) * b * n * fin
    return dims


def _dcgru_layer_impl(x, x_off: int, h0, p, p_batched: int, wg, bg, wc, bc, lengths, x_planes, n: int, h: int, m: int,
                      act: int, save: bool, want_hsel: bool, basis=None, pack=None, spack=None):
    """x: (T + x_off, B, N, Fin) — x_off = 1 when x is the `hext` of the layer below (its slot 0 is that layer's
    initial state), whose `hpl` output is then passed as x_planes.  Returns hext (T+1, B, N*H) (slot 0 = initial
    state, slot t+1 = h_t), hsel (B, N*H) = h at t = lengths-1 (T-1 without lengths) and the tensors the backward
    needs: [xtm, pack, planes, rs, us, cs, rhs, hpl, rhpl, spack] (numel-0 placeholders where nothing is kept).
    basis (`spectral_basis` of the ONE symmetric support all clips share; p_batched must be 0): the hoisted x-part runs in the
    eigenbasis of the support; `planes` is then the node-major transformed input (N, Sp, Fin) and spack the per-frequency packs.
    pack / spack: the packs of this cell made ahead (`pack_cells`: all layers in one launch); None: packed here."""
    lib = _lib.get_lib()
    if x.dim() != 4 or x.shape[2] != n:
        raise RuntimeError(f"inputs have shape {tuple(x.shape)}, expected (T, B, num_nodes={n}, input_dim)")
    t_len, b, fin = x.shape[0] - x_off, x.shape[1], x.shape[3]
    if not lib.query("eeg_dcrnn_supported", n, h, fin, m):
        raise RuntimeError("eeg_gnn_ssl_amd: " + lib.last_error())
    _check_polys(p, p_batched, b, n, m)
    if lengths is not None:
        _check_lengths(lengths, b)
    if h0 is not None and h0.numel() != b * n * h:
        raise RuntimeError(f"initial_hidden_state has shape {tuple(h0.shape)}, expected ({b}, {n * h})")
    ready = x_planes is not None
    spec = basis is not None
    dims = _layer_dims(t_len, b, n, h, fin, m, act, p_batched, ready)
    if spec:
        _check(lib, basis, "basis")
        sp_rows = int(lib.query("eeg_dcrnn_spectral_rows", t_len * b))
        if ready:      # x_planes = the layer below's U^T h (N, B + Sp, Fin): rows B.. of every frequency are this layer's transformed input
            dims.x_plane_stride = (b + sp_rows) * fin
        if p_batched or basis.numel() != _spectral_basis_floats(n) or not lib.query("eeg_dcrnn_spectral_ok", ctypes.byref(dims), 0):
            raise RuntimeError("dcgru_layer: the spectral form needs one shared support (p_batched = 0) and a shape "
                               "eeg_dcrnn_spectral_ok accepts")
        if ready and tuple(x_planes.shape) != (n, b + sp_rows, fin):
            raise RuntimeError(f"x_planes has shape {tuple(x_planes.shape)}, expected {(n, b + sp_rows, fin)} (the layer below's U^T h)")
    empty = _new((0,), p)
    # a transposed view of a contiguous batch-major (B,T,N,Fin) tensor (what model.py:253 produces) is consumed
    # as it is: the diffusion kernel emits the time-major copy as a by-product
    xsrc, xtm = None, None
    bm = 0
    if not x.is_contiguous() and x_off == 0 and not ready and x.transpose(0, 1).is_contiguous():
        bm = 2 if spec else lib.query("eeg_dcrnn_batch_major_ok", ctypes.byref(dims))   # (the spectral node mix reads any row order)
    if bm:
        xsrc = x.transpose(0, 1)
        _check(lib, xsrc, "inputs")
        if bm == 2:                       # no copy at all: diffusion kernel and GEMMs read (b,t) rows through a map
            dims.x_batch_major = 1
            xk = xsrc
        else:
            xtm = _new((t_len, b, n, fin), x)
            xk = xtm
    else:
        if not x.is_contiguous():
            xtm = x.contiguous()
            x = xtm
        _check(lib, x, "inputs")
        xk = x[x_off:] if x_off else x
    if h0 is not None:
        h0 = h0.contiguous()
        _check(lib, h0, "initial_hidden_state")
    _check(lib, p, "P")
    pack_given, spack_given = pack is not None, spack is not None
    if pack is None:
        pack = torch.ops.eeg_dcrnn.pack_cell(wg, bg, wc, bc, fin, h, m)
    elif pack.numel() < _pack_base_floats(fin, h, m):
        raise RuntimeError(f"dcgru_layer: pack has {pack.numel()} floats, a cell of input_dim={fin}, num_units={h}, num_matrices={m} needs "
                           f"{_pack_base_floats(fin, h, m)}")
    _set_pack3(dims, pack, fin, h, m)
    s = t_len * b
    if spec:
        if spack is None:
            spack = torch.ops.eeg_dcrnn.pack_cell_spectral(wg, wc, basis, fin, h, m, n)
        elif spack.numel() != lib.query("eeg_dcrnn_spectral_pack_floats", fin, h, m, n):
            raise RuntimeError(f"dcgru_layer: spack has {spack.numel()} floats, expected {lib.query('eeg_dcrnn_spectral_pack_floats', fin, h, m, n)}")
        dims.spectral, dims.spack = basis.data_ptr(), spack.data_ptr()
        if ready:
            _check(lib, x_planes, "x_planes")
            planes, planes_ptr = _new((0,), p), x_planes.data_ptr() + 4 * b * fin
        else:
            planes = _new((n, sp_rows, fin), x)
            planes_ptr = planes.data_ptr()
    elif ready:
        _check(lib, x_planes, "x_planes")
        if tuple(x_planes.shape) != (m - 1, t_len + 1, b, n, fin):
            raise RuntimeError(f"x_planes has shape {tuple(x_planes.shape)}, expected {(m - 1, t_len + 1, b, n, fin)}")
        planes, planes_ptr = _new((0,), p), x_planes.data_ptr() + 4 * b * n * fin      # (its own placeholder: outputs must not alias)
    else:
        planes = _new((m - 1, s, n, fin), x)
        planes_ptr = planes.data_ptr()
    hext = _new((t_len + 1, b, n * h), x)
    if save:
        rs, us, cs, rhs = (_new((t_len, b, n * h), x) for _ in range(4))
        if spec:       # the by-products in the eigenbasis: U^T h_slot (slots 0..T; the next layer's transformed input) and U^T (r*h_{t-1})
            hpl, rhpl = _new((n, b + sp_rows, h), x), _new((n, sp_rows, h), x)
        else:
            hpl, rhpl = (_new((m - 1, t_len + 1, b, n, h), x) for _ in range(2))
    else:
        rs = us = cs = rhs = hpl = rhpl = None
    ws = _new((lib.query("eeg_dcrnn_layer_fwd_ws_floats", ctypes.byref(dims)),), x)
    lib.call("eeg_dcrnn_layer_fwd", ctypes.byref(dims), _p(xsrc if xsrc is not None else xk), _p(xtm) if (xsrc is not None and bm != 2) else None,
             _p(h0), _p(p), _p(pack), planes_ptr, _p(hext), _p(rs), _p(us), _p(cs), _p(rhs), _p(hpl), _p(rhpl), _p(ws), _stream(x))
    if lengths is not None:
        lengths = lengths.to(device=x.device, dtype=torch.int64).contiguous()
        hsel = _new((b, n * h), x)
        lib.call("eeg_dcrnn_gather_last", _p(hext[1:]), _p(lengths), t_len, b, n * h, _p(hsel), _stream(x))
    elif want_hsel:
        hsel = hext[t_len].clone()
    else:
        hsel = _new((0,), x)           # nobody reads the final state of this layer (a lower layer of the classification model)
    if not save:
        return hext, hsel, []
    # outputs must not alias each other or an input: packs handed in are kept by the autograd context from the INPUTS
    return hext, hsel, [xtm if xtm is not None else empty, _new((0,), p) if pack_given else pack, planes, rs, us, cs, rhs, hpl, rhpl,
                        _new((0,), p) if (spack_given or not spec) else spack]


def _dcgru_layer_fake(x, x_off, h0, p, p_batched, wg, bg, wc, bc, lengths, x_planes, n, h, m, act, save, want_hsel, basis=None,
                      pack=None, spack=None):
    t_len, b, fin = x.shape[0] - x_off, x.shape[1], x.shape[3]
    ne = lambda *shape: x.new_empty(shape)   # noqa: E731
    hext, hsel = ne(t_len + 1, b, n * h), (ne(b, n * h) if (want_hsel or lengths is not None) else ne(0))
    if not save:
        return hext, hsel, []
    # a non-contiguous x is copied time-major (kept for the backward) -- except the transposed view of a contiguous batch-major
    # tensor at a layer without handed-over planes where the kernels read it through a row map: exactly when the library's host-side
    # predicate eeg_dcrnn_batch_major_ok says 2 (1: the diffusion kernel emits the time-major copy, 0: torch copies; both keep xtm)
    zero_copy = False
    spec = basis is not None
    if not x.is_contiguous() and x_off == 0 and x_planes is None and x.transpose(0, 1).is_contiguous():
        dims = _layer_dims(t_len, b, n, h, fin, m, act, p_batched, False)
        zero_copy = spec or _lib.get_lib().query("eeg_dcrnn_batch_major_ok", ctypes.byref(dims)) == 2
    xtm = ne(t_len, b, n, fin) if (not x.is_contiguous() and not zero_copy) else ne(0)
    planes = ne(0) if x_planes is not None else ne(m - 1, t_len * b, n, fin)
    spack_in, spack = spack, ne(0)
    byp = [ne(m - 1, t_len + 1, b, n, h) for _ in range(2)]
    if spec:
        lib = _lib.get_lib()
        sp_rows = int(lib.query("eeg_dcrnn_spectral_rows", t_len * b))
        if x_planes is None:
            planes = ne(n, sp_rows, fin)
        if spack_in is None:
            spack = ne(int(lib.query("eeg_dcrnn_spectral_pack_floats", fin, h, m, n)))
        byp = [ne(n, b + sp_rows, h), ne(n, sp_rows, h)]
    return hext, hsel, [xtm, ne(0) if pack is not None else ne(_pack_floats(fin, h, m)), planes] + [ne(t_len, b, n * h) for _ in range(4)] + byp + [spack]


def _dcgru_layer_bwd_impl(d_hext, d_hsel, x, x_off: int, p, p_batched: int, pack, planes, x_planes, hext, rs, us, cs, rhs,
                          hpl, rhpl, lengths, has_h0: bool, n: int, h: int, m: int, act: int, need_dx: bool, need_dh0: bool,
                          dwg, dbg, dwc, dbc, basis=None, spack=None):
    """BPTT + all parameter gradients of one layer.  d_hext (T+1,B,N*H) w.r.t. hext (slot 0 is ignored), d_hsel
    (B,N*H) w.r.t. hsel.  dwg/dbg/dwc/dbc are overwritten.  Returns (dx (T + x_off, B, N, Fin) with the step
    gradients in slots x_off.., dh0 (B,N*H)) — numel-0 tensors where not requested."""
    lib = _lib.get_lib()
    t_len, b, fin = hext.shape[0] - 1, hext.shape[1], x.shape[3]
    ready = x_planes is not None
    dims = _layer_dims(t_len, b, n, h, fin, m, act, p_batched, ready)
    if not x.is_contiguous():             # the forward consumed the batch-major input through the row map (saved as the view)
        if not x.transpose(0, 1).is_contiguous():
            raise RuntimeError("dcgru_layer_bwd: x must be contiguous or the transposed view of a batch-major tensor")
        dims.x_batch_major = 1
    planes_ptr = x_planes.data_ptr() + 4 * b * n * fin if ready else planes.data_ptr()
    _set_pack3(dims, pack, fin, h, m)
    if basis is not None:                 # the forward ran the spectral form: `planes` is its node-major transformed input
        if ready:                         # ... handed over by the layer below: rows B.. of every frequency of its U^T h (N, B + Sp, Fin)
            dims.x_plane_stride = x_planes.shape[1] * fin
            planes_ptr = x_planes.data_ptr() + 4 * b * fin
        if spack is None or not lib.query("eeg_dcrnn_spectral_ok", ctypes.byref(dims), 1 if need_dx else 0):
            raise RuntimeError("dcgru_layer_bwd: the spectral form does not cover this call (input gradient of a layer whose "
                               "input width is not 64?)")
        dims.spectral, dims.spack = basis.data_ptr(), spack.data_ptr()
    state = b * n * h
    if d_hext is not None:
        d_hext = d_hext.contiguous()
        _check(lib, d_hext, "grad of hext")
    if d_hsel is not None:
        d_hsel = d_hsel.contiguous()
    d_hseq_ptr = ctypes.c_void_p(d_hext.data_ptr() + 4 * state) if d_hext is not None else None
    d_at_end = d_hsel if lengths is None else None
    d_at_len = d_hsel if lengths is not None else None
    xk_ptr = ctypes.c_void_p(x.data_ptr() + 4 * x_off * b * n * fin)
    dx = _new((t_len + x_off, b, n, fin), hext) if need_dx else _new((0,), hext)
    if need_dx and x_off:
        # x is the `hext` of the layer below: its slot 0 is that layer's INITIAL state, which this layer never reads.  The
        # kernels fill slots x_off.. only; the gradient of slot 0 is exactly zero (and must be a defined value: it flows
        # through autograd accumulation, hooks and anomaly detection)
        lib.call("eeg_dcrnn_zero", _p(dx), 4 * x_off * b * n * fin, _stream(dx))
    dx_ptr = ctypes.c_void_p(dx.data_ptr() + 4 * x_off * b * n * fin) if need_dx else None
    dh0 = _new((b, n * h), hext) if (need_dh0 and has_h0) else _new((0,), hext)
    ws = _new((lib.query("eeg_dcrnn_layer_bwd_ws_floats", ctypes.byref(dims), 1 if need_dx else 0),), hext)
    lib.call("eeg_dcrnn_layer_bwd", ctypes.byref(dims), xk_ptr, _p(p), _p(pack), planes_ptr, _p(hext), _p(rs),
             _p(us), _p(cs), _p(rhs), _p(_none_if_empty(hpl)), _p(_none_if_empty(rhpl)), d_hseq_ptr, _p(d_at_end), _p(d_at_len), _p(lengths), dx_ptr,
             _p(dh0) if dh0.numel() else None, _p(dwg), _p(dbg), _p(dwc), _p(dbc), _p(ws), _stream(hext))
    return dx, dh0


def _dcgru_layer_bwd_fake(d_hext, d_hsel, x, x_off, p, p_batched, pack, planes, x_planes, hext, rs, us, cs, rhs, hpl, rhpl,
                          lengths, has_h0, n, h, m, act, need_dx, need_dh0, dwg, dbg, dwc, dbc, basis=None, spack=None):
    t_len, b, fin = hext.shape[0] - 1, hext.shape[1], x.shape[3]
    return (hext.new_empty((t_len + x_off, b, n, fin) if need_dx else (0,)),
            hext.new_empty((b, n * h) if (need_dh0 and has_h0) else (0,)))


_define("dcgru_layer",
        "(Tensor x, int x_off, Tensor? h0, Tensor P, int p_batched, Tensor wg, Tensor bg, Tensor wc, Tensor bc, Tensor? lengths, "
        "Tensor? x_planes, int n, int h, int m, int act, bool save, bool want_hsel, Tensor? basis, Tensor? pack=None, Tensor? spack=None) -> "
        "(Tensor hext, Tensor hsel, Tensor[] saved)",
        _dcgru_layer_impl, _dcgru_layer_fake)
_define("dcgru_layer_bwd",
        "(Tensor? d_hext, Tensor? d_hsel, Tensor x, int x_off, Tensor P, int p_batched, Tensor pack, Tensor planes, Tensor? x_planes, "
        "Tensor hext, Tensor rs, Tensor us, Tensor cs, Tensor rhs, Tensor hpl, Tensor rhpl, Tensor? lengths, bool has_h0, int n, int h, "
        "int m, int act, bool need_dx, bool need_dh0, Tensor(a!) dwg, Tensor(b!) dbg, Tensor(c!) dwc, Tensor(d!) dbc, "
        "Tensor? basis, Tensor? spack) -> (Tensor, Tensor)",
        _dcgru_layer_bwd_impl, _dcgru_layer_bwd_fake)


def _dcgru_layer_setup(ctx, inputs, output):
    (x, x_off, h0, p, p_batched, wg, bg, wc, bc, lengths, x_planes, n, h, m, act, save, want_hsel, basis, _pack, _spack) = inputs
    hext, _, saved = output
    ctx.set_materialize_grads(False)
    ctx.saved_ok = bool(save) and len(saved) == 10
    ctx.meta = (x_off, p_batched, n, h, m, act, h0 is not None)
    ctx.params = (wg, bg, wc, bc)
    if ctx.saved_ok:
        xtm = saved[0]
        lens = None if lengths is None else lengths.to(device=x.device, dtype=torch.int64).contiguous()
        kept = list(saved[1:])
        if _pack is not None:
            kept[0] = _pack
        if _spack is not None and basis is not None:
            kept[8] = _spack
        ctx.save_for_backward(xtm if xtm.numel() else x, p, x_planes, lens, basis, hext, *kept)


def _dcgru_layer_backward(ctx, d_hext, d_hsel, d_saved):
    if not ctx.saved_ok:
        raise RuntimeError("eeg_dcrnn::dcgru_layer was run with save=False: nothing was kept for the backward pass")
    xk, p, x_planes, lens, basis, hext, pack, planes, rs, us, cs, rhs, hpl, rhpl, spack = ctx.saved_tensors
    x_off, p_batched, n, h, m, act, has_h0 = ctx.meta
    need_dx, need_dh0 = ctx.needs_input_grad[0], has_h0 and ctx.needs_input_grad[2]
    fin = xk.shape[3]
    rows = (fin + h) * m
    shapes = ((rows, 2 * h), (2 * h,), (rows, h), (h,))
    sunk = [GradSink.take(q) for q in ctx.params]           # written in place -> nothing for autograd to add
    bufs = [t if t is not None else _new(sh, hext) for t, sh in zip(sunk, shapes)]
    dx, dh0 = torch.ops.eeg_dcrnn.dcgru_layer_bwd(d_hext, d_hsel, xk, x_off, p, p_batched, pack, planes, x_planes, hext,
                                                  rs, us, cs, rhs, hpl, rhpl, lens, has_h0, n, h, m, act, need_dx, need_dh0, *bufs,
                                                  basis, spack if basis is not None else None)
    ret = [None if t is not None else g for t, g in zip(sunk, bufs)]
    grads = (dx if need_dx else None, None, dh0 if need_dh0 else None, None, None, *ret, None, None, None, None, None, None, None, None,
             None, None, None)
    # trailing arguments a caller left at their defaults (pack, spack) are not inputs of THIS call: one gradient slot per given input
    return grads[:len(ctx.needs_input_grad)]


torch.library.register_autograd(f"{NS}::dcgru_layer", _dcgru_layer_backward, setup_context=_dcgru_layer_setup, lib=_libdef)


# =============================================================================================
# decoder (model.py:160-204) as one operator
# =============================================================================================
def _dec_dims(meta, dropout_p=0.0, teacher_on_device=False):
    t_len, b, n, h, dout, m, n_layers, act, p_batched = meta
    return DecoderDims(t_len, b, n, h, dout, m, n_layers, act, p_batched, float(dropout_p), 1 if teacher_on_device else 0)


def _teacher_arg(lib, teacher, teacher_dev, t_len):
    """the `teacher` argument of eeg_dcrnn_decoder_fwd / _bwd: (void* value, keep-alive object, flags on the device?)"""
    if teacher_dev is not None:
        _check(lib, teacher_dev, "teacher_dev", torch.int32)
        if teacher_dev.numel() != t_len:
            raise RuntimeError(f"teacher_dev: expected int32[{t_len}], got {tuple(teacher_dev.shape)}")
        return _p(teacher_dev), teacher_dev, True
    if len(teacher) > 0 and any(teacher):
        arr = (ctypes.c_int32 * t_len)(*[1 if teacher[i] else 0 for i in range(t_len)])
        return ctypes.cast(arr, ctypes.c_void_p), arr, False
    return None, None, False


def make_rng_state(device, stream_id: int = 0) -> torch.Tensor:
    """State {seed, offset} (int64[2]) of the Philox4x32-10 generator behind the fused dropout masks (csrc/common.h).  The seed
    is a function of `torch.initial_seed()` (so `torch.manual_seed` governs it like it governs `nn.Dropout`) and of `stream_id`
    (the classification head and the decoder use different ones), drawn from a DEDICATED generator: the global CPU generator
    is not advanced.  Every forward call that drops advances the offset ON THE DEVICE (a replayed HIP graph keeps drawing
    fresh masks)."""
    rank = 0
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank = torch.distributed.get_rank()           # same-seed ranks draw different masks (as their data shards differ)
    g = torch.Generator().manual_seed((torch.initial_seed() + 0x9E3779B97F4A7C15 * int(stream_id) + 0xD1B54A32D192ED03 * rank) % (1 << 63))
    seed = int(torch.randint(0, 2 ** 62, (1,), generator=g).item())
    return torch.tensor([seed, 0], dtype=torch.int64, device=device)


def _check_rng(lib, t, name):
    _check(lib, t, name, torch.int64)
    if t.numel() != 2:
        raise RuntimeError(f"{name}: expected int64[2] {{seed, offset}}, got {tuple(t.shape)}")


def _dcgru_decoder_impl(targets, h0, p, p_batched: int, wg0, bg0, wc0, bc0, wg1, bg1, wc1, bc1, wp, bp, teacher: List[int],
                        t_len: int, n: int, h: int, dout: int, m: int, n_layers: int, act: int, dropout_p: float, rng_used,
                        teacher_dev=None):
    """T autoregressive steps through L cells + the projection.  teacher: T ints (1 = feed targets[t] to step t+1,
    all 0 = fully autoregressive) -- or teacher_dev: the same flags as a DEVICE int32[T] tensor (`teacher_flags_`), read by the
    persistent kernel when it starts (graph-replayable curriculum learning); dropout_p > 0: nn.Dropout in front of the projection (model.py:191), masks from the
    {seed, offset} pair `rng_used` that `rng_take_` handed out for T*B*N*H/4 counters.  Returns out (T,B,N*Dout) and
    [saved, pack0, pack1]."""
    lib = _lib.get_lib()
    if dropout_p > 0:
        if rng_used is None:
            raise RuntimeError("dcgru_decoder: dropout_p > 0 needs rng_used (ops.rng_take)")
        _check_rng(lib, rng_used, "rng_used")
    for fin in (dout, h):
        if not lib.query("eeg_dcrnn_supported", n, h, fin, m):
            raise RuntimeError("eeg_gnn_ssl_amd: " + lib.last_error())
    h0 = h0.contiguous()
    b = h0.shape[1]
    wp, bp = wp.detach().contiguous(), bp.detach().contiguous()
    for t, nm in ((h0, "initial_hidden_state"), (p, "P"), (wp, "projection_layer.weight"), (bp, "projection_layer.bias")):
        _check(lib, t, nm)
    if tuple(wp.shape) != (dout, h) or tuple(bp.shape) != (dout,):
        raise RuntimeError(f"projection_layer shapes {tuple(wp.shape)}, {tuple(bp.shape)} do not match ({dout}, {h})")
    if tuple(h0.shape) != (n_layers, b, n * h):
        raise RuntimeError(f"initial_hidden_state has shape {tuple(h0.shape)}, expected ({n_layers}, {b}, {n * h})")
    _check_polys(p, p_batched, b, n, m)
    if targets is not None and targets.numel() != t_len * b * n * dout:
        raise RuntimeError(f"decoder inputs have shape {tuple(targets.shape)}, expected ({t_len}, {b}, {n}*{dout}) "
                           f"(T, B, num_nodes * output_dim)")
    tf_ptr, tf_keep, tf_dev = _teacher_arg(lib, teacher, teacher_dev, t_len)
    use_tf = tf_ptr is not None
    if use_tf:
        if targets is None:
            raise RuntimeError("teacher forcing needs the target sequence")
        targets = targets.contiguous()
        _check(lib, targets, "inputs (teacher-forcing targets)")
    pack0 = torch.ops.eeg_dcrnn.pack_cell(wg0, bg0, wc0, bc0, dout, h, m)
    pack1 = torch.ops.eeg_dcrnn.pack_cell(wg1, bg1, wc1, bc1, h, h, m) if n_layers > 1 else _new((0,), h0)
    packs = [pack0] + [pack1] * (n_layers - 1)
    dims = _dec_dims((t_len, b, n, h, dout, m, n_layers, act, p_batched), dropout_p, tf_dev)
    out = _new((t_len, b, n * dout), h0)
    saved = _new((lib.query("eeg_dcrnn_decoder_saved_floats", ctypes.byref(dims)),), h0)
    ws = _new((lib.query("eeg_dcrnn_decoder_fwd_ws_floats", ctypes.byref(dims)),), h0)
    pk_arr = (ctypes.c_void_p * n_layers)(*[q.data_ptr() for q in packs])
    lib.call("eeg_dcrnn_decoder_fwd", ctypes.byref(dims), _p(targets) if use_tf else None, tf_ptr, _p(h0), _p(p), pk_arr,
             _p(wp), _p(bp), _p(rng_used) if dropout_p > 0 else None, _p(out), _p(saved), _p(ws), _stream(h0))
    return out, [saved, pack0, pack1]


def _dcgru_decoder_bwd_impl(d_out, p, p_batched: int, saved, pack0, pack1, wp, teacher: List[int], t_len: int, n: int, h: int,
                            dout: int, m: int, n_layers: int, act: int, dropout_p: float, rng_used,
                            dwg0, dbg0, dwc0, dbc0, dwg1, dbg1, dwc1, dbc1, dwp, dbp, teacher_dev=None):
    lib = _lib.get_lib()
    d_out = d_out.contiguous()
    _check(lib, d_out, "grad of the decoder output")
    b = d_out.shape[1]
    tf_ptr, tf_keep, tf_dev = _teacher_arg(lib, teacher, teacher_dev, t_len)
    dims = _dec_dims((t_len, b, n, h, dout, m, n_layers, act, p_batched), dropout_p, tf_dev)
    if dropout_p > 0:
        _check_rng(lib, rng_used, "rng_used")
    packs = [pack0] + [pack1] * (n_layers - 1)
    g0, g1 = (dwg0, dbg0, dwc0, dbc0), (dwg1, dbg1, dwc1, dbc1)
    dh0 = _new((n_layers, b, n * h), saved)
    ws = _new((lib.query("eeg_dcrnn_decoder_bwd_ws_floats", ctypes.byref(dims)),), saved)
    arr = lambda k: (ctypes.c_void_p * n_layers)(*[(g0 if l == 0 else g1)[k].data_ptr() for l in range(n_layers)])  # noqa: E731
    pk_arr = (ctypes.c_void_p * n_layers)(*[q.data_ptr() for q in packs])
    lib.call("eeg_dcrnn_decoder_bwd", ctypes.byref(dims), tf_ptr, _p(p), pk_arr, _p(wp.detach().contiguous()), _p(saved), _p(d_out),
             _p(rng_used) if dropout_p > 0 else None, _p(dh0), arr(0), arr(1), arr(2), arr(3), _p(dwp), _p(dbp), _p(ws), _stream(saved))
    return dh0


def _dec_saved_floats_fake(h0, t_len, n, h, dout, m, n_layers):
    return h0.new_empty((1,))        # opaque block: only the device implementation knows its size


_define("dcgru_decoder",
        "(Tensor? targets, Tensor h0, Tensor P, int p_batched, Tensor wg0, Tensor bg0, Tensor wc0, Tensor bc0, Tensor? wg1, Tensor? bg1, "
        "Tensor? wc1, Tensor? bc1, Tensor wp, Tensor bp, int[] teacher, int t_len, int n, int h, int dout, int m, int n_layers, int act, "
        "float dropout_p, Tensor? rng_used, Tensor? teacher_dev) -> (Tensor out, Tensor[] saved)",
        _dcgru_decoder_impl,
        lambda targets, h0, p, p_batched, wg0, bg0, wc0, bc0, wg1, bg1, wc1, bc1, wp, bp, teacher, t_len, n, h, dout, m, n_layers, act,
        dropout_p, rng_used, teacher_dev=None:
        (h0.new_empty((t_len, h0.shape[1], n * dout)),
         [_dec_saved_floats_fake(h0, t_len, n, h, dout, m, n_layers), h0.new_empty((_pack_floats(dout, h, m),)),
          h0.new_empty((_pack_floats(h, h, m) if n_layers > 1 else 0,))]))
_define("dcgru_decoder_bwd",
        "(Tensor d_out, Tensor P, int p_batched, Tensor saved, Tensor pack0, Tensor pack1, Tensor wp, int[] teacher, int t_len, int n, "
        "int h, int dout, int m, int n_layers, int act, float dropout_p, Tensor? rng_used, Tensor(a!) dwg0, Tensor(b!) dbg0, "
        "Tensor(c!) dwc0, Tensor(d!) dbc0, Tensor(e!)? dwg1, Tensor(f!)? dbg1, Tensor(g!)? dwc1, Tensor(h!)? dbc1, Tensor(i!) dwp, "
        "Tensor(j!) dbp, Tensor? teacher_dev) -> Tensor",
        _dcgru_decoder_bwd_impl,
        lambda d_out, p, p_batched, saved, pack0, pack1, wp, teacher, t_len, n, h, dout, m, n_layers, act, dropout_p, rng_used, *grads:
        d_out.new_empty((n_layers, d_out.shape[1], n * h)))


def _dcgru_decoder_setup(ctx, inputs, output):
    (targets, h0, p, p_batched, wg0, bg0, wc0, bc0, wg1, bg1, wc1, bc1, wp, bp, teacher, t_len, n, h, dout, m, n_layers, act,
     dropout_p, rng_used, teacher_dev) = inputs
    _, saved = output
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(p, wp, rng_used, teacher_dev, *saved)
    ctx.params = (wg0, bg0, wc0, bc0, wg1, bg1, wc1, bc1, wp, bp)
    ctx.meta = (p_batched, list(teacher), t_len, n, h, dout, m, n_layers, act, float(dropout_p))


def _dcgru_decoder_backward(ctx, d_out, d_saved):
    p, wp, rng_used, teacher_dev, saved, pack0, pack1 = ctx.saved_tensors
    p_batched, teacher, t_len, n, h, dout, m, n_layers, act, dropout_p = ctx.meta
    if d_out is None:
        return (None,) * 25
    shapes = [None if q is None else tuple(q.shape) for q in ctx.params]
    sunk = [GradSink.take(q) if q is not None else None for q in ctx.params]
    bufs = [t if t is not None else (_new(sh, saved) if sh is not None else None) for t, sh in zip(sunk, shapes)]
    dh0 = torch.ops.eeg_dcrnn.dcgru_decoder_bwd(d_out, p, p_batched, saved, pack0, pack1, wp, teacher, t_len, n, h, dout, m,
                                                n_layers, act, dropout_p, rng_used if dropout_p > 0 else None, *bufs, teacher_dev)
    ret = [None if (t is not None or g is None) else g for t, g in zip(sunk, bufs)]   # sunk: already in the caller's buffer
    return (None, dh0, None, None, *ret, None, None, None, None, None, None, None, None, None, None, None)


torch.library.register_autograd(f"{NS}::dcgru_decoder", _dcgru_decoder_backward, setup_context=_dcgru_decoder_setup, lib=_libdef)


# =============================================================================================
# heads, gather, graph construction, featurisation
# =============================================================================================
def _cls_head_impl(z, w, bias, dropout_p: float, rng_used):
    lib = _lib.get_lib()
    z, w, bias = z.contiguous(), w.detach().contiguous(), bias.detach().contiguous()
    for t, nm in ((z, "last_out"), (w, "fc.weight"), (bias, "fc.bias")):
        _check(lib, t, nm)
    if z.dim() != 3:
        raise RuntimeError(f"cls_head: last_out has shape {tuple(z.shape)}, expected (B, num_nodes, rnn_units)")
    b, n, h = z.shape
    c = w.shape[0]
    if tuple(w.shape) != (c, h) or bias.numel() != c:
        raise RuntimeError(f"cls_head: fc.weight {tuple(w.shape)} / fc.bias {tuple(bias.shape)} do not match rnn_units={h}")
    drop = dropout_p > 0
    if drop:
        if rng_used is None:
            raise RuntimeError("cls_head: dropout_p > 0 needs rng_used (ops.rng_take)")
        _check_rng(lib, rng_used, "rng_used")
    logits = _new((b, c), z)
    arg = _new((b, c), z, torch.int32)
    lib.call("eeg_dcrnn_cls_head_fwd", _p(z), _p(w), _p(bias), b, n, h, c, float(dropout_p), _p(rng_used) if drop else None,
             _p(logits), _p(arg), _stream(z))
    return logits, arg


def _rng_take_impl(rng_state, groups: int):
    """{seed, offset} of the device generator -> a fresh int64[2] tensor; the state's offset advances by `groups` counters on the
    stream (so a replayed HIP graph keeps drawing fresh masks)"""
    lib = _lib.get_lib()
    _check_rng(lib, rng_state, "rng_state")
    used = torch.empty_like(rng_state)
    lib.call("eeg_dcrnn_rng_take", _p(rng_state), int(groups), _p(used), _stream(rng_state))
    return used


def _cls_head_bwd_impl(z, w, dlogits, arg, dropout_p: float, rng_used, dw, db):
    lib = _lib.get_lib()
    z, w, dlogits = z.contiguous(), w.detach().contiguous(), dlogits.contiguous()
    b, n, h = z.shape
    c = w.shape[0]
    if tuple(w.shape) != (c, h) or tuple(dlogits.shape) != (b, c) or tuple(arg.shape) != (b, c) or arg.dtype != torch.int32 \
            or tuple(dw.shape) != (c, h) or db.numel() != c:
        raise RuntimeError(f"cls_head_bwd: operands do not fit: z {tuple(z.shape)}, w {tuple(w.shape)}, dlogits {tuple(dlogits.shape)}, "
                           f"arg {tuple(arg.shape)} {arg.dtype}, dw {tuple(dw.shape)}, db {tuple(db.shape)}")
    dz = torch.empty_like(z)
    lib.call("eeg_dcrnn_cls_head_bwd", _p(z), _p(w), _p(dlogits), _p(arg), b, n, h, w.shape[0], float(dropout_p),
             _p(rng_used) if dropout_p > 0 else None, _p(dz), _p(dw), _p(db), _stream(z))
    return dz


def _dropout_mask_impl(rng_used, n: int, dropout_p: float):
    """the mask x 1/(1-p) factors the fused kernels apply to elements 0..n-1 of a dropped tensor for the {seed, offset} pair a
    forward call reported (tests: handed to the oracle)"""
    lib = _lib.get_lib()
    _check_rng(lib, rng_used, "rng_used")
    mask = torch.empty((n,), dtype=torch.float32, device=rng_used.device)
    lib.call("eeg_dcrnn_dropout_mask", _p(rng_used), n, float(dropout_p), _p(mask), _stream(rng_used))
    return mask


_define("rng_take_", "(Tensor(a!) rng_state, int groups) -> Tensor", _rng_take_impl, lambda rng_state, groups: torch.empty_like(rng_state))
_define("cls_head", "(Tensor z, Tensor w, Tensor bias, float dropout_p, Tensor? rng_used) -> (Tensor logits, Tensor arg)",
        _cls_head_impl,
        lambda z, w, bias, dropout_p, rng_used: (z.new_empty((z.shape[0], w.shape[0])), z.new_empty((z.shape[0], w.shape[0]), dtype=torch.int32)))
_define("cls_head_bwd", "(Tensor z, Tensor w, Tensor dlogits, Tensor arg, float dropout_p, Tensor? rng_used, Tensor(a!) dw, Tensor(b!) db) -> Tensor",
        _cls_head_bwd_impl, lambda z, w, dlogits, arg, dropout_p, rng_used, dw, db: torch.empty_like(z))
_define("dropout_mask", "(Tensor rng_used, int n, float dropout_p) -> Tensor", _dropout_mask_impl,
        lambda rng_used, n, dropout_p: rng_used.new_empty((n,), dtype=torch.float32))


def _cls_head_setup(ctx, inputs, output):
    z, w, bias, dropout_p, rng_used = inputs
    ctx.save_for_backward(z, w, output[1], rng_used)
    ctx.params = (w, bias)
    ctx.dropout_p = float(dropout_p)
    ctx.set_materialize_grads(False)


def _cls_head_backward(ctx, dlogits, _darg):
    z, w, arg, rng_used = ctx.saved_tensors
    if dlogits is None:
        return None, None, None, None, None
    sunk = [GradSink.take(q) for q in ctx.params]
    dw = sunk[0] if sunk[0] is not None else torch.empty_like(w)
    db = sunk[1] if sunk[1] is not None else _new((w.shape[0],), z)
    dz = torch.ops.eeg_dcrnn.cls_head_bwd(z, w, dlogits, arg, ctx.dropout_p, rng_used if ctx.dropout_p > 0 else None, dw, db)
    return dz, (None if sunk[0] is not None else dw), (None if sunk[1] is not None else db), None, None


torch.library.register_autograd(f"{NS}::cls_head", _cls_head_backward, setup_context=_cls_head_setup, lib=_libdef)


def _cls_head_loss_impl(z, w, bias, targets, kind: int, dropout_p: float, rng_used, dw, db):
    """head forward + criterion + the head's backward for one optimisation step (eeg_dcrnn_cls_head_loss): returns
    (loss (1,), logits (B,C), arg (B,C) int32, dlogits (B,C), dz (B,N,H)); dw (C,H) / db (C) are overwritten."""
    lib = _lib.get_lib()
    z, w, bias = z.contiguous(), w.detach().contiguous(), bias.detach().contiguous()
    for t, nm in ((z, "head input"), (w, "fc.weight"), (bias, "fc.bias"), (dw, "fc.weight gradient"), (db, "fc.bias gradient")):
        _check(lib, t, nm)
    if z.dim() != 3:
        raise RuntimeError(f"cls_head_loss: head input has shape {tuple(z.shape)}, expected (B, num_nodes, rnn_units)")
    b, n, h = z.shape
    c = w.shape[0]
    if tuple(w.shape) != (c, h) or bias.numel() != c or tuple(dw.shape) != (c, h) or db.numel() != c or not dw.is_contiguous() or not db.is_contiguous():
        raise RuntimeError(f"cls_head_loss: fc operands {tuple(w.shape)}, {tuple(bias.shape)}, gradients {tuple(dw.shape)}, {tuple(db.shape)} do not "
                           f"fit a head input of {h} units")
    if kind == 0:
        tg = targets.to(torch.float32).contiguous().view(-1)
        _check(lib, tg, "targets")
    else:
        tg = targets.to(torch.int64).contiguous().view(-1)
        _check(lib, tg, "targets", torch.int64)
    if tg.numel() != b:
        raise RuntimeError(f"cls_head_loss: {tg.numel()} targets for {b} clips")
    drop = dropout_p > 0
    if drop:
        if rng_used is None:
            raise RuntimeError("cls_head_loss: dropout_p > 0 needs rng_used (ops.rng_take)")
        _check_rng(lib, rng_used, "rng_used")
    loss, logits, arg = _new((1,), z), _new((b, c), z), _new((b, c), z, torch.int32)
    dlogits, dz = _new((b, c), z), torch.empty_like(z)
    ws = _new((lib.query("eeg_dcrnn_cls_head_loss_ws_floats", b, h, c),), z)
    lib.call("eeg_dcrnn_cls_head_loss", _p(z), _p(w), _p(bias), _p(tg), int(kind), b, n, h, c, float(dropout_p), _p(rng_used) if drop else None,
             _p(logits), _p(arg), _p(dlogits), _p(dz), _p(dw), _p(db), _p(loss), _p(ws), _stream(z))
    return loss, logits, arg, dlogits, dz


def _cls_head_loss_fake(z, w, bias, targets, kind, dropout_p, rng_used, dw, db):
    b, c = z.shape[0], w.shape[0]
    return (z.new_empty((1,)), z.new_empty((b, c)), z.new_empty((b, c), dtype=torch.int32), z.new_empty((b, c)), torch.empty_like(z))


_define("cls_head_loss", "(Tensor z, Tensor w, Tensor bias, Tensor targets, int kind, float dropout_p, Tensor? rng_used, Tensor(a!) dw, Tensor(b!) db) "
        "-> (Tensor, Tensor, Tensor, Tensor, Tensor)", _cls_head_loss_impl, _cls_head_loss_fake)


def _gather_last_impl(htop, lengths):
    lib = _lib.get_lib()
    htop = htop.contiguous()
    _check(lib, htop, "output")
    t_len, b, d = htop.shape
    _check_lengths(lengths, b)
    lengths = lengths.to(device=htop.device, dtype=torch.int64).contiguous()
    out = _new((b, d), htop)
    lib.call("eeg_dcrnn_gather_last", _p(htop), _p(lengths), t_len, b, d, _p(out), _stream(htop))
    return out


_define("gather_last", "(Tensor htop, Tensor lengths) -> Tensor", _gather_last_impl,
        lambda htop, lengths: htop.new_empty((htop.shape[1], htop.shape[2])))


def _corr_graph_impl(x, top_k: int):
    lib = _lib.get_lib()
    x = x.contiguous()
    _check(lib, x, "clips")
    if x.dim() != 4:
        raise RuntimeError(f"clips must be (B,T,N,D), got {tuple(x.shape)}")
    b, t_len, n, d = x.shape
    adj, s1, s2 = (_new((b, n, n), x) for _ in range(3))
    ws = _new((lib.query("eeg_dcrnn_corr_graph_ws_floats", b, t_len),), x)
    lib.call("eeg_dcrnn_corr_graph", _p(x), b, t_len, n, d, int(top_k), _p(adj), _p(s1), _p(s2), _p(ws), _stream(x))
    return adj, s1, s2


_define("corr_graph", "(Tensor x, int top_k) -> (Tensor adj, Tensor s1, Tensor s2)", _corr_graph_impl,
        lambda x, top_k: tuple(x.new_empty((x.shape[0], x.shape[2], x.shape[2])) for _ in range(3)))


def _fft_features_impl(raw, window: int, mean: float, std: float, standardise: bool, perm, log_scale):
    lib = _lib.get_lib()
    raw = raw.contiguous()
    _check(lib, raw, "raw signals")
    if raw.dim() != 3 or raw.shape[2] % window != 0:
        raise RuntimeError(f"raw signals must be (B, N, T*{window}), got {tuple(raw.shape)}")
    b, n, total = raw.shape
    t_len = total // window
    feat_raw = _new((b, t_len, n, window // 2), raw)
    feat_std = torch.empty_like(feat_raw) if standardise else _new((0,), raw)
    if perm is not None:
        if perm.numel() != b * n:
            raise RuntimeError(f"fft_features: perm has shape {tuple(perm.shape)}, expected ({b}, {n}) source channels")
        # Every row must be a permutation of 0..N-1 (feat_raw is written at the SOURCE slot: a row that is not one would leave slots
        # of the torch.empty allocation unwritten).  A host tensor is checked here (no device sync involved); a device tensor is the
        # output of eeg_dcrnn_augment_draw (a permutation by construction) or the caller's responsibility.
        if lib.is_device_build and perm.device.type == "cpu" and not bool((torch.sort(perm.reshape(b, n).to(torch.int64), dim=1).values
                                                    == torch.arange(n, dtype=torch.int64)).all()):
            raise RuntimeError("fft_features: every row of perm must be a permutation of 0..N-1")
        perm = perm.to(device=raw.device, dtype=torch.int32).contiguous()
    if log_scale is not None and log_scale.numel() != b:
        raise RuntimeError(f"fft_features: log_scale has {log_scale.numel()} entries for {b} clips")
    if log_scale is not None:
        log_scale = log_scale.to(device=raw.device, dtype=torch.float32).contiguous()
    lib.call("eeg_dcrnn_fft_features", _p(raw), b, n, t_len, window, _p(perm), _p(log_scale), float(mean), float(std),
             _p(feat_raw), _p(feat_std) if standardise else None, _stream(raw))
    return feat_raw, feat_std


def _fft_features_fake(raw, window, mean, std, standardise, perm, log_scale):
    shape = (raw.shape[0], raw.shape[2] // window, raw.shape[1], window // 2)
    return raw.new_empty(shape), raw.new_empty(shape if standardise else (0,))


_define("fft_features", "(Tensor raw, int window, float mean, float std, bool standardise, Tensor? perm, Tensor? log_scale) -> (Tensor, Tensor)",
        _fft_features_impl, _fft_features_fake)


# =============================================================================================
# losses that seed backward, optimiser tail
# =============================================================================================
def _bce_logits_impl(logits, y):
    lib = _lib.get_lib()
    x = logits.contiguous().view(-1)
    yy = y.to(torch.float32).contiguous().view(-1)
    _check(lib, x, "logits")
    _check(lib, yy, "targets")
    if yy.numel() != x.numel():
        raise RuntimeError(f"bce_logits: {yy.numel()} targets for {x.numel()} logits")
    loss, dx = _new((1,), x), torch.empty_like(x)
    lib.call("eeg_dcrnn_bce_logits", _p(x), _p(yy), x.numel(), _p(loss), _p(dx), _stream(x))
    return loss[0], dx.view(logits.shape)


def _ce_logits_impl(logits, y):
    lib = _lib.get_lib()
    x = logits.contiguous()
    yy = y.to(torch.int64).contiguous()
    _check(lib, x, "logits")
    _check(lib, yy, "targets", torch.int64)
    if x.dim() != 2 or yy.numel() != x.shape[0]:
        raise RuntimeError(f"ce_logits: logits {tuple(x.shape)} need shape (B, C) and {yy.numel()} targets B entries")
    loss, dx = _new((1,), x), torch.empty_like(x)
    lib.call("eeg_dcrnn_ce_logits", _p(x), _p(yy), x.shape[0], x.shape[1], _p(loss), _p(dx), _stream(x))
    return loss[0], dx


def _masked_loss_impl(pred, y, use_scaler: bool, mean: float, std: float, mask_val: float, kind: int):
    lib = _lib.get_lib()
    pr = pred.contiguous()
    t = y.to(torch.float32).contiguous()
    # the kernels read 16-byte vectors: a contiguous VIEW with an odd element offset (pred.view(-1)[1:]) is not aligned
    if pr.data_ptr() % 16:
        pr = pr.clone()
    if t.data_ptr() % 16:
        t = t.clone()
    _check(lib, pr, "y_predicted")
    _check(lib, t, "y_true")
    if pr.shape != t.shape:
        raise RuntimeError(f"y_predicted {tuple(pr.shape)} and y_true {tuple(t.shape)} differ in shape")
    loss, dp = _new((1,), pr), torch.empty_like(pr)
    ws = _new((lib.query("eeg_dcrnn_masked_loss_ws_floats"),), pr)
    lib.call("eeg_dcrnn_masked_loss", _p(pr), _p(t), pr.numel(), 1 if use_scaler else 0, float(mean), float(std),
             float(mask_val), int(kind), _p(loss), _p(dp), _p(ws), _stream(pr))
    return loss[0], dp


_loss_fake = lambda x, *a: (x.new_empty(()), torch.empty_like(x))   # noqa: E731
_define("bce_logits", "(Tensor logits, Tensor y) -> (Tensor loss, Tensor dlogits)", _bce_logits_impl, _loss_fake)
_define("ce_logits", "(Tensor logits, Tensor y) -> (Tensor loss, Tensor dlogits)", _ce_logits_impl, _loss_fake)
_define("masked_loss", "(Tensor pred, Tensor y, bool use_scaler, float mean, float std, float mask_val, int kind) -> (Tensor loss, Tensor dpred)",
        _masked_loss_impl, _loss_fake)


def _loss_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.n_in = len(inputs)


def _loss_backward(ctx, dloss, _dgrad):
    (dx,) = ctx.saved_tensors
    return (dx * dloss,) + (None,) * (ctx.n_in - 1)


for _name in ("bce_logits", "ce_logits", "masked_loss"):
    torch.library.register_autograd(f"{NS}::{_name}", _loss_backward, setup_context=_loss_setup, lib=_libdef)


def _clip_adam_impl(params, grads, exp_avg, exp_avg_sq, step: int, lr: float, beta1: float, beta2: float, eps: float,
                    weight_decay: float, max_norm: float, grad_scale: float, ws, norm_out):
    lib = _lib.get_lib()
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _check(lib, t, nm)
    lib.call("eeg_dcrnn_clip_adam", _p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), params.numel(), float(max_norm),
             float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale),
             _p(ws), _p(norm_out), _stream(params))


_define("clip_adam_",
        "(Tensor(a!) params, Tensor(b!) grads, Tensor(c!) exp_avg, Tensor(d!) exp_avg_sq, int step, float lr, float beta1, float beta2, "
        "float eps, float weight_decay, float max_norm, float grad_scale, Tensor(e!) ws, Tensor(f!)? norm_out) -> ()",
        _clip_adam_impl, lambda *a: None)


def _clip_adam_dev_impl(params, grads, exp_avg, exp_avg_sq, step_dev, lr_dev, beta1: float, beta2: float, eps: float,
                        weight_decay: float, max_norm: float, grad_scale: float, ws, norm_out):
    lib = _lib.get_lib()
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (lr_dev, "lr_dev")):
        _check(lib, t, nm)
    _check(lib, step_dev, "step_dev", torch.int32)
    lib.call("eeg_dcrnn_clip_adam_dev", _p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), params.numel(), float(max_norm),
             _p(lr_dev), float(beta1), float(beta2), float(eps), float(weight_decay), _p(step_dev), float(grad_scale),
             _p(ws), _p(norm_out), _stream(params))


_define("clip_adam_dev_",
        "(Tensor(a!) params, Tensor(b!) grads, Tensor(c!) exp_avg, Tensor(d!) exp_avg_sq, Tensor(g!) step_dev, Tensor lr_dev, float beta1, "
        "float beta2, float eps, float weight_decay, float max_norm, float grad_scale, Tensor(e!) ws, Tensor(f!)? norm_out) -> ()",
        _clip_adam_dev_impl, lambda *a: None)


def _teacher_flags_impl(rng_state, samples_seen, increment: int, cl_decay_steps: float, t_len: int):
    lib = _lib.get_lib()
    _check_rng(lib, rng_state, "rng_state")
    _check(lib, samples_seen, "samples_seen", torch.int64)
    flags = _new((t_len,), rng_state, torch.int32)
    lib.call("eeg_dcrnn_teacher_flags", _p(rng_state), _p(samples_seen), int(increment), float(cl_decay_steps), int(t_len),
             _p(flags), _stream(rng_state))
    return flags


_define("teacher_flags_", "(Tensor(a!) rng_state, Tensor(b!) samples_seen, int increment, float cl_decay_steps, int t_len) -> Tensor",
        _teacher_flags_impl, lambda rng_state, samples_seen, increment, cl_decay_steps, t_len: rng_state.new_empty((t_len,), dtype=torch.int32))


def _augment_draw_impl(rng_state, batch: int, swap_perm, plain_supports, reflected_supports):
    lib = _lib.get_lib()
    _check_rng(lib, rng_state, "rng_state")
    _check(lib, swap_perm, "swap_perm", torch.int32)
    n = swap_perm.numel()
    if batch < 1 or n < 1:
        raise RuntimeError(f"augment_draw: batch={batch}, num_nodes={n}")
    used = _rng_take_impl(rng_state, int(batch))
    flags = _new((batch,), rng_state, torch.int32)
    perm = _new((batch, n), rng_state, torch.int32)
    log_scale = _new((batch,), rng_state, torch.float32)
    nsup, s_out = 0, _new((0,), rng_state, torch.float32)
    if plain_supports is not None or reflected_supports is not None:
        if plain_supports is None or reflected_supports is None:
            raise RuntimeError("augment_draw: the per-clip supports need BOTH the plain and the reflected set")
        plain_supports, reflected_supports = plain_supports.contiguous(), reflected_supports.contiguous()
        _check(lib, plain_supports, "plain_supports")
        _check(lib, reflected_supports, "reflected_supports")
        if plain_supports.dim() != 3 or tuple(plain_supports.shape[1:]) != (n, n) or plain_supports.shape != reflected_supports.shape:
            raise RuntimeError(f"augment_draw: supports must be two (n_supports, {n}, {n}) tensors, got {tuple(plain_supports.shape)} and "
                               f"{tuple(reflected_supports.shape)}")
        nsup = plain_supports.shape[0]
        s_out = _new((nsup, batch, n, n), rng_state, torch.float32)
    lib.call("eeg_dcrnn_augment_draw", _p(used), int(batch), int(n), _p(swap_perm), _p(flags), _p(perm), _p(log_scale),
             _p(plain_supports) if nsup else None, _p(reflected_supports) if nsup else None, int(nsup), _p(s_out) if nsup else None,
             _stream(rng_state))
    return flags, perm, log_scale, s_out


def _augment_draw_fake(rng_state, batch, swap_perm, plain_supports, reflected_supports):
    n = swap_perm.numel()
    so = (0,) if plain_supports is None else (plain_supports.shape[0], batch, n, n)
    return (rng_state.new_empty((batch,), dtype=torch.int32), rng_state.new_empty((batch, n), dtype=torch.int32),
            rng_state.new_empty((batch,), dtype=torch.float32), rng_state.new_empty(so, dtype=torch.float32))


_define("augment_draw_", "(Tensor(a!) rng_state, int batch, Tensor swap_perm, Tensor? plain_supports, Tensor? reflected_supports) -> "
        "(Tensor, Tensor, Tensor, Tensor)", _augment_draw_impl, _augment_draw_fake)


# =============================================================================================
# Python conveniences used by model/, utils.py and train_step.py
# =============================================================================================
# diagnostics: how often a layer took its input hop planes from the recurrent kernel of the layer below / ran its hoisted x-part
# in the eigenbasis of a shared symmetric support
hop_plane_handovers = 0
spectral_layer_calls = 0


def hop_polys(supports: Sequence[torch.Tensor], max_diffusion_step: int, batch: int) -> Tuple[torch.Tensor, int]:
    """Hop-polynomial matrices of the clip graphs (cell.py:83-93 incl. quirk Q1).

    supports: list of (N,N) or (B,N,N) tensors (torch.matmul broadcast semantics of cell.py:85).
    Returns (P (G, M-1, N, N), p_batched) with G = B if any support is batched else 1."""
    sups = list(supports)
    if len(sups) == 0:
        raise RuntimeError("hop_polys: empty supports list")
    flag = 1 if any(s.dim() == 3 for s in sups) else 0
    if flag or any(s.requires_grad for s in sups):
        return torch.ops.eeg_dcrnn.hop_polys(sups, int(max_diffusion_step), int(batch)), flag
    # graphs shared by all clips (2-D supports: the distance graph, one constant tensor for a whole run): the polynomials are a pure
    # function of the supports -- built once per supports tensors (identity + version), not once per forward
    import weakref
    key = (int(max_diffusion_step),) + tuple((s.data_ptr(), s._version, str(s.device), tuple(s.shape), s.dtype) for s in sups)
    hit = _polys_cache.get(key)
    if hit is not None and all(r() is s for r, s in zip(hit[0], sups)):
        return hit[1], 0
    out = torch.ops.eeg_dcrnn.hop_polys(sups, int(max_diffusion_step), int(batch))
    if not (out.is_cuda and torch.cuda.is_current_stream_capturing()):      # (a tensor born inside a capture lives in the graph's pool)
        if len(_polys_cache) > 64:
            for k in [k for k, v in _polys_cache.items() if any(r() is None for r in v[0])]:
                del _polys_cache[k]
        _polys_cache[key] = ([weakref.ref(s) for s in sups], out)
    return out, 0


# (max_diffusion_step, identity + version of every support) -> (weakrefs, P): see hop_polys
_polys_cache = {}


# id of a batched supports tensor -> (weakref, its 2-D form or None)
_shared_cache = {}


def collapse_shared_supports(supports):
    """[S (B,N,N)] whose B clips all carry the SAME graph -> [S[0] (N,N)]: the 2-D form declares the graph shared, which is what
    lets the encoder run the spectral form (`shared_spectral_basis`).  The reference's trainers always pass batched supports, even
    for the one distance graph, so `TrainStep` asks here.  Anything else (several supports, per-clip graphs, already
    2-D, spectral mode off) is returned unchanged.  The comparison reads one flag back from the device -- once per supports
    tensor (cached on its identity and version); under stream capture an unseen tensor is returned unchanged."""
    if not SPECTRAL_MODE or supports is None or len(supports) != 1 or not torch.is_tensor(supports[0]) or supports[0].dim() != 3:
        return supports
    import weakref
    sup = supports[0]
    key = (sup.data_ptr(), sup._version, str(sup.device), tuple(sup.shape), sup.dtype)
    hit = _shared_cache.get(key)
    if hit is not None and hit[0]() is sup:
        return supports if hit[1] is None else [hit[1]]
    if sup.is_cuda and torch.cuda.is_current_stream_capturing():
        return supports
    if len(_shared_cache) > 64:
        for k in [k for k, v in _shared_cache.items() if v[0]() is None]:
            del _shared_cache[k]
    same = sup.shape[0] >= 1 and bool((sup == sup[0:1]).all().item())
    flat = sup[0].to(torch.float32).clone() if same else None
    _shared_cache[key] = (weakref.ref(sup), flat)
    return supports if flat is None else [flat]


def fft_features(raw: torch.Tensor, window: int = 200, mean: Optional[float] = None, std: Optional[float] = None,
                 perm: Optional[torch.Tensor] = None, log_scale: Optional[torch.Tensor] = None):
    """Input featurisation on the device: raw (B,N,T*window) resampled signals ->
    (feat_raw (B,T,N,window/2) log|FFT| per 1-s step, feat_std = the standardised (and optionally
    augmented) model input or None when mean/std are not given).

    Replaces `computeFFT` per step (data_utils.py:13-35, dataloader_detection.py:57-71), the reflection /
    amplitude-jitter augmentation (perm (B,N) int32 source channel per node, log_scale (B);
    dataloader_detection.py:233-256) and `StandardScaler.transform` (utils.py:393-428)."""
    std_on = mean is not None
    fr, fs = torch.ops.eeg_dcrnn.fft_features(raw, int(window), float(mean) if std_on else 0.0,
                                              float(std) if std is not None else 1.0, std_on, perm, log_scale)
    return fr, (fs if std_on else None)


def correlation_supports(x: torch.Tensor, top_k: int = 3, return_adj: bool = False):
    """Per-clip correlation graph -> [S1, S2] dual random-walk supports, on the device.

    x (B,T,N,D) clips (the model input).  Replaces the DataLoader-side `_get_indiv_graphs` +
    `keep_topk` + `_compute_supports('dual_random_walk')` (dataloader_detection.py:258-307,335-354).
    Returns [S1 (B,N,N), S2 (B,N,N)] (and the sparsified adjacency (B,N,N) if return_adj)."""
    adj, s1, s2 = torch.ops.eeg_dcrnn.corr_graph(x, int(top_k))
    return ([s1, s2], adj) if return_adj else [s1, s2]


def pack_cell(wg, bg, wc, bc, fin: int, h: int, m: int) -> torch.Tensor:
    """Reference-layout cell parameters -> MFMA-fragment-ordered device block (kernels_pack.h)."""
    return torch.ops.eeg_dcrnn.pack_cell(wg, bg, wc, bc, fin, h, m)


def diffusion_hops(x: torch.Tensor, p: torch.Tensor, p_batched: int, batch: int) -> torch.Tensor:
    """The diffusion step alone: x (S,N,F) -> (M-1,S,N,F) hop planes P_m x (micro-benchmark entry;
    north_star's HBM-bound kernel)."""
    return torch.ops.eeg_dcrnn.diffusion_hops(x, p, int(p_batched), int(batch))


def dconv(x, p, p_batched, weight, biases):
    """DiffusionGraphConv.forward (cell.py:66-118) on x (B,N,F): HIP diffusion + fp32-MFMA GEMM with
    the reference-layout weight ((F*M), O); differentiable w.r.t. x, weight and biases."""
    return torch.ops.eeg_dcrnn.dconv(x, p, int(p_batched), weight, biases)


class LayerOut:
    """what one layer hands on: hext (T+1,B,N*H) with hseq = hext[1:], hsel (B,N*H) and — when the backward
    pass will run — the hop planes hpl (M-1,T+1,B,N,H) whose slots 1..T are the next layer's input planes"""
    __slots__ = ("hext", "hsel", "hpl")

    def __init__(self, hext, hsel, hpl):
        self.hext, self.hsel, self.hpl = hext, hsel, hpl

    @property
    def hseq(self):
        return self.hext[1:]


def pack_encoder_cells(cells, basis, num_nodes):
    """(packs, spacks) of the cells of an encoder from ONE launch (`pack_cells`), or None where the cells pack themselves (a single
    layer, more than four, the opt-in bf16 split).  spacks is a list of None without a basis."""
    cells = list(cells)
    if len(cells) < 2 or len(cells) > 4 or GEMM_MODE != 0:
        return None
    h, m = cells[0]._num_units, cells[0].num_matrices
    if basis is not None and h != 64:
        basis = None
    # (detached: the packs are a re-layout the layer operators consume next to the parameters themselves, whose gradients come from
    #  the layer operators' backward)
    out = torch.ops.eeg_dcrnn.pack_cells([c.dconv_gate.weight.detach() for c in cells], [c.dconv_gate.biases.detach() for c in cells],
                                         [c.dconv_candidate.weight.detach() for c in cells], [c.dconv_candidate.biases.detach() for c in cells],
                                         h, m, basis, int(num_nodes))
    k = len(cells)
    return out[:k], (out[k:] if basis is not None else [None] * k)


def dcgru_layer_ex(x, x_off, h0, p, p_batched, wg, bg, wc, bc, n, h, m, activation="tanh", lengths=None, x_planes=None,
                   want_hsel=True, basis=None, pack=None, spack=None) -> LayerOut:
    """One DCGRU layer over x (T + x_off, B, N, Fin).  x_off = 1 / x_planes: x is the `hext` of the layer below and
    x_planes its `hpl` (the layer then skips its own diffusion pass).  want_hsel=False: the caller does not read the layer's
    final state (hsel comes back empty; saves one copy per step).  basis (`shared_spectral_basis`): the hoisted x-part of the
    layer runs in the eigenbasis of the shared symmetric support where the kernels cover the shape (else the general path)."""
    act = ACT_CODES.get(activation, 1)
    save = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, h0, wg, bg, wc, bc))
    if basis is not None:
        need_dx = 1 if (save and x.requires_grad) else 0
        dims = _layer_dims(x.shape[0] - x_off, x.shape[1], n, h, x.shape[3], m, act, int(p_batched), False)
        if p_batched or not _lib.get_lib().query("eeg_dcrnn_spectral_ok", ctypes.byref(dims), need_dx):
            basis = None
        else:
            if x_planes is not None and x_planes.dim() != 3:
                x_planes = None            # (hop planes of a layer that took the general path)
            global spectral_layer_calls
            spectral_layer_calls += 1
    if basis is None and x_planes is not None and x_planes.dim() != 5:
        x_planes = None                    # (the U^T h of a spectral layer is of no use to the general path)
    if x_planes is not None:
        global hop_plane_handovers
        hop_plane_handovers += 1
    hext, hsel, saved = torch.ops.eeg_dcrnn.dcgru_layer(x, int(x_off), h0, p, int(p_batched), wg, bg, wc, bc, lengths, x_planes,
                                                        n, h, m, act, save, bool(want_hsel), basis, pack,
                                                        spack if basis is not None else None)
    return LayerOut(hext, hsel, saved[7].detach() if (save and saved[7].numel()) else None)


# (data_ptr, version, device, shape) of a support tensor -> (weakref to it, basis or None): the eigenbasis is a pure function of the
# support, the distance graph is one constant tensor for a whole run, and its residual has to be READ once (a device-to-host
# synchronisation) to know that the support is symmetric -- neither belongs into a training step.
_basis_cache = {}


def shared_spectral_basis(supports, max_diffusion_step: int):
    """The eigenbasis block of the support when the spectral form applies: exactly ONE support, given as a 2-D (N,N) tensor
    (= declared shared by all clips; batched supports always take the general path), at least one diffusion step, and
    |S - U diag(lam) U^T| at rounding level (S symmetric: filter_type "laplacian" on an undirected graph).  Else None.
    The first call for a support synchronises once to read the residual; under stream capture an unseen support gets None."""
    if not SPECTRAL_MODE or max_diffusion_step < 1 or len(supports) != 1:
        return None
    sup = supports[0]
    if sup.dim() != 2 or sup.shape[0] != sup.shape[1] or sup.shape[0] < 2 or sup.shape[0] > 32:
        return None
    import weakref
    key = (sup.data_ptr(), sup._version, str(sup.device), tuple(sup.shape), sup.dtype)
    hit = _basis_cache.get(key)
    if hit is not None and hit[0]() is sup:
        return hit[1]
    if sup.is_cuda and torch.cuda.is_current_stream_capturing():
        return None
    if len(_basis_cache) > 64:
        for k in [k for k, v in _basis_cache.items() if v[0]() is None]:
            del _basis_cache[k]
    basis = torch.ops.eeg_dcrnn.spectral_basis(sup)
    n = sup.shape[0]
    info = basis[n * n + 256:n * n + 259].tolist()        # residual, off-diagonal left, max |S|  (one synchronisation per support)
    ok = info[0] <= SPECTRAL_TOL * max(info[2], 1e-30) and info[1] <= 1e-9 * max(info[2], 1e-30)
    _basis_cache[key] = (weakref.ref(sup), basis if ok else None)
    return basis if ok else None


def dcgru_layer(x, h0, p, p_batched, wg, bg, wc, bc, n, h, m, activation="tanh", lengths=None):
    """Run one DCGRU layer over x (T,B,N,Fin).  Returns (hseq (T,B,N*H), hsel (B,N*H))."""
    out = dcgru_layer_ex(x, 0, h0, p, p_batched, wg, bg, wc, bc, n, h, m, activation, lengths)
    return out.hseq, out.hsel


def dcgru_decoder(targets, h0, p, p_batched, first_cell, shared_cell, wp, bp, n, h, dout, m, n_layers,
                  activation="tanh", teacher=None, dropout_p=0.0, rng_state=None, return_rng_used=False):
    """Run the whole decoder: returns (T,B,N*Dout).  first_cell / shared_cell = (wg, bg, wc, bc).  dropout_p > 0: nn.Dropout
    in front of the projection, masks from the device generator rng_state (make_rng_state; advanced in place)."""
    act = ACT_CODES.get(activation, 1)
    t_len = targets.shape[0]
    sc = shared_cell if shared_cell is not None else (None, None, None, None)
    # teacher: a sequence of T host flags, or a DEVICE int32[T] tensor (ops.teacher_flags: drawn on the stream, graph-replayable)
    tf_dev = teacher if torch.is_tensor(teacher) else None
    use_tf = tf_dev is not None or (teacher is not None and any(teacher))
    tf = [1 if v else 0 for v in teacher] if (use_tf and tf_dev is None) else [0] * t_len     # never an empty list: pytree leaf of the autograd glue
    used = rng_take(rng_state, t_len * h0.shape[1] * n * h // 4) if dropout_p > 0 else None
    out, _ = torch.ops.eeg_dcrnn.dcgru_decoder(targets if use_tf else None, h0, p, int(p_batched), *first_cell, *sc, wp, bp, tf,
                                               int(t_len), n, h, dout, m, n_layers, act, float(dropout_p), used, tf_dev)
    return (out, used) if return_rng_used else out


def cls_head(z, w, bias, dropout_p=0.0, rng_state=None, return_rng_used=False):
    """model.py:267-270: dropout (training) -> relu -> per-node Linear(H->C) -> max over nodes, one launch; the dropout mask
    comes from the device generator rng_state (make_rng_state) and is recomputed, not stored, in the backward."""
    used = rng_take(rng_state, z.numel() // 4) if dropout_p > 0 else None
    logits, _ = torch.ops.eeg_dcrnn.cls_head(z, w, bias, float(dropout_p), used)
    return (logits, used) if return_rng_used else logits


def cls_head_loss(z, fc_weight, fc_bias, targets, task="detection", dropout_p=0.0, rng_state=None):
    """The tail of a supervised optimisation step behind the encoder (model.py:267-270 + train.py:203-206,266-272) in two launches:
    dropout -> relu -> fc -> max over nodes, the criterion (task "detection": BCE-with-logits, "classification": cross-entropy) and
    their backward.  z (B,N,H) = the top layer's state at len-1, DETACHED from autograd: the gradients of fc go straight into
    `fc_weight.grad` / `fc_bias.grad` (written in place inside `with GradSink`, else accumulated), and the caller seeds the encoder's
    backward with the returned dz (`last.backward(dz)`).  Returns (loss (), logits (B,C), dz (B,N,H))."""
    used = rng_take(rng_state, z.numel() // 4) if dropout_p > 0 else None
    sunk = [GradSink.take(q) for q in (fc_weight, fc_bias)]
    dw = sunk[0] if sunk[0] is not None else torch.empty_like(fc_weight)
    db = sunk[1] if sunk[1] is not None else torch.empty_like(fc_bias)
    loss, logits, _arg, _dl, dz = torch.ops.eeg_dcrnn.cls_head_loss(z.detach(), fc_weight.detach(), fc_bias.detach(), targets,
                                                                    0 if task == "detection" else 1, float(dropout_p), used, dw, db)
    for q, buf, sk in ((fc_weight, dw, sunk[0]), (fc_bias, db, sunk[1])):
        if sk is None and q.requires_grad:
            q.grad = buf if q.grad is None else q.grad + buf
    return loss[0], logits, dz


def rng_take(rng_state, groups):
    """hand out the generator's current {seed, offset} pair for `groups` Philox counters and advance the device state"""
    if rng_state is None:
        raise RuntimeError("dropout_p > 0 needs a device generator state (ops.make_rng_state)")
    return torch.ops.eeg_dcrnn.rng_take_(rng_state, int(groups))


def dropout_mask(rng_used, n, dropout_p):
    """the keep-mask x 1/(1-p) factors of elements 0..n-1 for a forward call's {seed, offset} pair (tests)"""
    return torch.ops.eeg_dcrnn.dropout_mask(rng_used, int(n), float(dropout_p))


def gather_last(htop: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """utils.last_relevant_pytorch on a time-major (T,B,D) tensor, no host sync (forward only)."""
    return torch.ops.eeg_dcrnn.gather_last(htop, lengths)


def masked_regression_loss(y_predicted, y_true, mean=None, std=None, loss_fn="mae", mask_val=0.0):
    """utils.compute_regression_loss (utils.py:431-495).  Only the exact string 'mae' selects the MAE
    (utils.py:489-495); anything else is the masked RMSE."""
    scaled = mean is not None
    return torch.ops.eeg_dcrnn.masked_loss(y_predicted, y_true, scaled, float(mean) if scaled else 0.0,
                                           float(std) if scaled else 1.0, float(mask_val), 0 if loss_fn == "mae" else 1)[0]


def bce_with_logits(logits, y):
    """nn.BCEWithLogitsLoss() (mean): value and dlogits from one HIP launch (train.py:203-204)."""
    return torch.ops.eeg_dcrnn.bce_logits(logits, y)[0]


def cross_entropy(logits, y):
    """nn.CrossEntropyLoss() (mean) on (B,C) logits and int64 class targets (train.py:205-206)."""
    return torch.ops.eeg_dcrnn.ce_logits(logits, y)[0]


def decoder_is_persistent(t_len, batch, n, h, dout, m, n_layers) -> bool:
    """the persistent decoder kernels cover this shape (then device-resident teacher-forcing flags are available)"""
    lib = _lib.get_lib()
    dims = _dec_dims((int(t_len), int(batch), int(n), int(h), int(dout), int(m), int(n_layers), 0, 0))
    return bool(lib.query("eeg_dcrnn_decoder_is_persistent", ctypes.byref(dims)))


def teacher_flags(rng_state, samples_seen, increment, cl_decay_steps, t_len):
    """Scheduled-sampling flags of one decoder forward drawn ON THE DEVICE (model.py:194-200, utils.py:385-390): int32[t_len],
    flag t = u_t < k / (k + exp(samples_seen / k)); advances the generator and adds `increment` to samples_seen on the stream."""
    return torch.ops.eeg_dcrnn.teacher_flags_(rng_state, samples_seen, int(increment), float(cl_decay_steps), int(t_len))


def draw_augmentation(rng_state, batch, swap_perm, plain_supports=None, reflected_supports=None):
    """The reference's per-sample augmentation draws (dataloader_detection.py:233-256,384-389), ON THE DEVICE from a Philox
    generator state (`make_rng_state`; advanced on the stream: a captured step draws afresh at every replay): per clip a fair coin
    -- reflect along the midline or not -- and a scale factor uniform in [0.8, 1.2).
    swap_perm: int32 (N,) source channel of every node of a reflected clip (`utils.swap_permutation`).
    plain_supports / reflected_supports: lists (or stacked tensors) of the (N,N) supports of the distance graph and of its reflected
    partner (`utils.compute_supports` / `utils.reflected_supports`); given, the per-clip supports are returned as well.
    Returns (flags int32 (B,), perm int32 (B,N), log_scale float32 (B,), supports: list of (B,N,N) tensors or None) -- `perm` and
    `log_scale` are the operands of `fft_features`."""
    stack = lambda t: None if t is None else (t if torch.is_tensor(t) else torch.stack(list(t)))   # noqa: E731
    ps, rs = stack(plain_supports), stack(reflected_supports)
    if ps is not None and ps.dim() == 2:
        ps, rs = ps[None], rs[None]
    flags, perm, log_scale, s_out = torch.ops.eeg_dcrnn.augment_draw_(rng_state, int(batch), swap_perm, ps, rs)
    return flags, perm, log_scale, (None if ps is None else [s_out[i] for i in range(s_out.shape[0])])


def clip_adam_step_dev(params, grads, exp_avg, exp_avg_sq, step_dev, lr_dev, betas, eps, weight_decay, max_norm, grad_scale, ws,
                       norm_out=None):
    """clip_adam_step with the step counter (int32[1], incremented on the stream) and the learning rate (float32[1]) in device
    memory: capturable into a HIP graph together with the rest of the step."""
    torch.ops.eeg_dcrnn.clip_adam_dev_(params, grads, exp_avg, exp_avg_sq, step_dev, lr_dev, float(betas[0]), float(betas[1]),
                                       float(eps), float(weight_decay), float(max_norm), float(grad_scale), ws, norm_out)


def clip_adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay, max_norm, grad_scale, ws,
                   norm_out=None):
    """Fused clip_grad_norm_ + Adam (coupled L2) over flat fp32 buffers (train.py:273-275)."""
    torch.ops.eeg_dcrnn.clip_adam_(params, grads, exp_avg, exp_avg_sq, int(step), float(lr), float(betas[0]), float(betas[1]),
                                   float(eps), float(weight_decay), float(max_norm), float(grad_scale), ws, norm_out)

