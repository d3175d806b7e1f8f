"""torch-facing operators over the C ABI (include/eeg_dcrnn.h): tensors in, tensors out, autograd.

PyTorch is plumbing here — device memory, streams, autograd bookkeeping; all arithmetic of the
DCRNN path runs in the HIP kernels of libeeg_dcrnn_hip.so.  Every function raises if its
tensors are not on the GPU the library was built for: this package has no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import DecoderDims, LayerDims

ACT_CODES = {"tanh": 0, "relu": 1, None: 1}   # the reference maps anything but 'tanh' to relu (cell.py:146)


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


def num_matrices(filter_type: str, max_diffusion_step: int) -> int:
    """cell.py:35,151-158."""
    return (2 if filter_type == "dual_random_walk" else 1) * max_diffusion_step + 1


# ---------------------------------------------------------------------------------------------
# small memo so that per-step cell calls (decoder) do not rebuild hop polynomials / weight packs
# ---------------------------------------------------------------------------------------------
class _Memo:
    def __init__(self, cap=8):
        self.cap, self.items = cap, []

    @staticmethod
    def key(tensors, extra=()):
        return tuple((t.data_ptr(), t._version, tuple(t.shape), str(t.device)) for t in tensors) + tuple(extra)

    def get(self, k):
        for kk, v in self.items:
            if kk == k:
                return v
        return None

    def put(self, k, v):
        self.items.append((k, v))
        if len(self.items) > self.cap:
            self.items.pop(0)


_poly_memo = _Memo()
_pack_memo = _Memo(16)


class GradSink:
    """Lets the backward operators write parameter gradients straight into caller-owned buffers
    (TrainStep's flat gradient bucket) instead of returning fresh tensors that autograd then adds into
    `p.grad` with one small kernel per parameter.  Active only inside `with GradSink(params): ...`;
    a parameter that receives a second gradient in the same pass falls back to the returned-tensor path
    (autograd accumulates), so results never depend on the sink."""
    _active: Optional["GradSink"] = None

    def __init__(self, params: Sequence[torch.Tensor]):
        self.targets = {p.data_ptr(): p.grad for p in params if p.grad is not None}
        self.seen = set()

    def __enter__(self):
        GradSink._active = self
        self.seen.clear()
        return self

    def __exit__(self, *exc):
        GradSink._active = None
        return False

    @staticmethod
    def take(param: torch.Tensor) -> Optional[torch.Tensor]:
        """the buffer to write `param`'s gradient into, or None (-> allocate and return it to autograd)"""
        sink = GradSink._active
        if sink is None:
            return None
        key = param.data_ptr()
        tgt = sink.targets.get(key)
        if tgt is None or key in sink.seen or not tgt.is_contiguous() or tgt.shape != param.shape:
            return None
        sink.seen.add(key)
        return tgt


def new_forward_scope():
    """Drop the memoised hop polynomials / weight packs.  Called at the start of every encoder /
    decoder forward, so the memo only ever serves repeated cell calls INSIDE one forward (the
    decoder's time loop) and can never hand out packs of stale weights (e.g. after an optimiser
    that updates parameters through an aliased flat buffer, which does not bump `_version`)."""
    _poly_memo.items.clear()
    _pack_memo.items.clear()
    _hop_planes.clear()


# hop planes P_m h_t that a layer's recurrent kernel left behind (slots 1..T of its Hplanes), keyed by the
# data pointer of the hidden sequence they belong to: the next layer, fed that very tensor and the same
# hop polynomials, takes them as its input planes instead of diffusing again.  The entry keeps the hidden
# sequence's storage alive (no pointer reuse inside a scope); cleared by new_forward_scope().
_hop_planes: dict = {}
hop_plane_handovers = 0            # diagnostics: how often a layer took its input planes from the layer below
hop_plane_handover_enabled = True  # tests switch it off to compare against the separately diffused planes


def hop_polys(supports: Sequence[torch.Tensor], max_diffusion_step: int, batch: int) -> Tuple[torch.Tensor, int]:
    """Hop-polynomial matrices of the clip graphs (cell.py:83-93 incl. quirk Q1).

    supports: list of (N,N) or (B,N,N) tensors (torch.matmul broadcast semantics of cell.py:85).
    Returns (P (G, M-1, N, N), p_batched) with G = B if any support is batched else 1."""
    lib = _lib.get_lib()
    sups = list(supports)
    if len(sups) == 0:
        raise RuntimeError("hop_polys: empty supports list")
    k = _Memo.key(sups, (max_diffusion_step, batch))
    hit = _poly_memo.get(k)
    if hit is not None:
        return hit[0], hit[1]
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
    m1 = len(norm) * max_diffusion_step
    out = torch.empty((g, m1, n, n), dtype=torch.float32, device=norm[0].device)
    arr = (ctypes.c_void_p * len(norm))(*[s.data_ptr() for s in norm])
    lib.call("eeg_dcrnn_hop_polys", arr, len(norm), g, n, max_diffusion_step, _p(out), _stream(out))
    flag = 1 if batched else 0
    _poly_memo.put(k, (out, flag, norm, sups))   # keep the keyed tensors alive with the cache entry
    return out, flag


def fft_features(raw: torch.Tensor, window: int = 200, mean: Optional[float] = None, std: Optional[float] = None,
                 perm: Optional[torch.Tensor] = None, log_scale: Optional[torch.Tensor] = None):
    """Input featurisation on the device: raw (B,N,T*window) resampled signals ->
    (feat_raw (B,T,N,window/2) log|FFT| per 1-s step, feat_std = the standardised (and optionally
    augmented) model input or None when mean/std are not given).

    Replaces `computeFFT` per step (data_utils.py:13-35, dataloader_detection.py:57-71), the reflection /
    amplitude-jitter augmentation (perm (B,N) int32 source channel per node, log_scale (B);
    dataloader_detection.py:233-256) and `StandardScaler.transform` (utils.py:393-428)."""
    lib = _lib.get_lib()
    raw = raw.contiguous()
    _check(lib, raw, "raw signals")
    if raw.dim() != 3 or raw.shape[2] % window != 0:
        raise RuntimeError(f"raw signals must be (B, N, T*{window}), got {tuple(raw.shape)}")
    b, n, total = raw.shape
    t_len = total // window
    feat_raw = torch.empty((b, t_len, n, window // 2), dtype=torch.float32, device=raw.device)
    feat_std = torch.empty_like(feat_raw) if mean is not None else None
    if perm is not None:
        perm = perm.to(device=raw.device, dtype=torch.int32).contiguous()
    if log_scale is not None:
        log_scale = log_scale.to(device=raw.device, dtype=torch.float32).contiguous()
    lib.call("eeg_dcrnn_fft_features", _p(raw), b, n, t_len, window, _p(perm), _p(log_scale),
             float(mean) if mean is not None else 0.0, float(std) if std is not None else 1.0,
             _p(feat_raw), _p(feat_std), _stream(raw))
    return feat_raw, feat_std


def correlation_supports(x: torch.Tensor, top_k: int = 3, return_adj: bool = False):
    """Per-clip correlation graph -> [S1, S2] dual random-walk supports, on the device.

    x (B,T,N,D) clips (the model input).  Replaces the DataLoader-side `_get_indiv_graphs` +
    `keep_topk` + `_compute_supports('dual_random_walk')` (dataloader_detection.py:258-307,335-354).
    Returns [S1 (B,N,N), S2 (B,N,N)] (and the sparsified adjacency (B,N,N) if return_adj)."""
    lib = _lib.get_lib()
    x = x.contiguous()
    _check(lib, x, "clips")
    if x.dim() != 4:
        raise RuntimeError(f"clips must be (B,T,N,D), got {tuple(x.shape)}")
    b, t_len, n, d = x.shape
    s1 = torch.empty((b, n, n), dtype=torch.float32, device=x.device)
    s2 = torch.empty_like(s1)
    adj = torch.empty_like(s1) if return_adj else None
    ws = torch.empty(lib.query("eeg_dcrnn_corr_graph_ws_floats", b, t_len), dtype=torch.float32, device=x.device)
    lib.call("eeg_dcrnn_corr_graph", _p(x), b, t_len, n, d, int(top_k), _p(adj), _p(s1), _p(s2), _p(ws), _stream(x))
    return ([s1, s2], adj) if return_adj else [s1, s2]


def pack_cell(wg, bg, wc, bc, fin: int, h: int, m: int) -> torch.Tensor:
    """Reference-layout cell parameters -> MFMA-fragment-ordered device block (kernels_pack.h)."""
    lib = _lib.get_lib()
    tensors = [wg.detach(), bg.detach(), wc.detach(), bc.detach()]
    k = _Memo.key(tensors, (fin, h, m))
    hit = _pack_memo.get(k)
    if hit is not None:
        return hit
    for t, nm in zip(tensors, ("dconv_gate.weight", "dconv_gate.biases", "dconv_candidate.weight", "dconv_candidate.biases")):
        _check(lib, t, nm)
    rows = (fin + h) * m
    if tuple(wg.shape) != (rows, 2 * h) or tuple(wc.shape) != (rows, h) or tuple(bg.shape) != (2 * h,) or tuple(bc.shape) != (h,):
        raise RuntimeError(f"cell parameter shapes {tuple(wg.shape)}, {tuple(bg.shape)}, {tuple(wc.shape)}, {tuple(bc.shape)} "
                           f"do not match input_dim={fin}, num_units={h}, num_matrices={m}")
    n = lib.query("eeg_dcrnn_pack_floats", fin, h, m)
    pack = torch.empty(n, dtype=torch.float32, device=wg.device)
    lib.call("eeg_dcrnn_pack_cell", _p(tensors[0]), _p(tensors[1]), _p(tensors[2]), _p(tensors[3]), fin, h, m, _p(pack), _stream(pack))
    _pack_memo.put(k, pack)
    return pack


def diffusion_hops(x: torch.Tensor, p: torch.Tensor, p_batched: int, batch: int) -> torch.Tensor:
    """The diffusion step alone: x (S,N,F) -> (M-1,S,N,F) hop planes P_m x (micro-benchmark entry;
    north_star's HBM-bound kernel)."""
    lib = _lib.get_lib()
    _check(lib, x, "x")
    _check(lib, p, "P")
    s, n, f = x.shape
    m = p.shape[1] + 1
    out = torch.empty((m - 1, s, n, f), dtype=torch.float32, device=x.device)
    lib.call("eeg_dcrnn_diffuse_fwd", _p(x), _p(p), p_batched, s, batch, n, f, m, _p(out), _stream(x))
    return out


def dconv_forward(x, p, p_batched, weight, biases):
    """DiffusionGraphConv.forward (cell.py:66-118) on x (B,N,F): HIP diffusion + fp32-MFMA GEMM with
    the reference-layout weight ((F*M), O).  Forward only."""
    lib = _lib.get_lib()
    x = x.contiguous()
    w = weight.detach().contiguous()
    bvec = biases.detach().contiguous()
    for t, nm in ((x, "inputs_and_state"), (p, "P"), (w, "weight"), (bvec, "biases")):
        _check(lib, t, nm)
    b, n, f = x.shape
    m, o = p.shape[1] + 1, w.shape[1]
    if w.shape[0] != f * m:
        raise RuntimeError(f"weight has {w.shape[0]} rows, expected (input_dim+hid_dim)*num_matrices = {f * m}")
    out = torch.empty((b, n, o), dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.query("eeg_dcrnn_dconv_fwd_ws_floats", b, n, f, m, o), dtype=torch.float32, device=x.device)
    lib.call("eeg_dcrnn_dconv_fwd", _p(x), _p(p), p_batched, b, n, f, m, _p(w), _p(bvec), o, _p(out), _p(ws), _stream(x))
    return out


class _DCGRULayerFn(torch.autograd.Function):
    """One DCGRU layer over a whole sequence (the `for t` loop of model.py:93-96 around
    DCGRUCell.forward, cell.py:182-210), fwd + explicit BPTT backward in HIP.

    inputs : x (T,B,N,Fin), h0 (B,N*H) or None, P, wg, bg, wc, bc, lengths (int64 (B,) or None)
    outputs: hseq (T,B,N*H), hsel (B,N*H) = h at t = lengths-1 (or T-1 when lengths is None)."""

    @staticmethod
    def forward(ctx, x, h0, p, wg, bg, wc, bc, lengths, p_batched, n, h, m, act, track):
        lib = _lib.get_lib()
        t_len, b = x.shape[0], x.shape[1]
        fin = x.shape[3]
        dims = LayerDims(t_len, b, n, h, fin, m, act, p_batched)
        # input = the hidden sequence of the layer below, whose kernel already formed P_m x: take its planes
        ready = _hop_planes.get(x.data_ptr()) if (x.is_contiguous() and hop_plane_handover_enabled) else None
        if ready is not None and (ready["key"] != (t_len, b, n, fin, m, p.data_ptr(), p_batched)
                                  or ready["hext"]._version != ready["version"]):
            ready = None
        if ready is not None:
            global hop_plane_handovers
            hop_plane_handovers += 1
            dims.x_planes_ready = 1
            dims.x_plane_stride = (t_len + 1) * b * n * fin
        # a transposed view of a contiguous batch-major (B,T,N,Fin) tensor (what model.py:253 produces) is
        # consumed as it is: the diffusion kernel emits the time-major copy as a by-product
        xsrc, xtm = None, None
        if not x.is_contiguous() and x.transpose(0, 1).is_contiguous() and lib.query("eeg_dcrnn_batch_major_ok", ctypes.byref(dims)):
            xsrc = x.transpose(0, 1)
            _check(lib, xsrc, "inputs")
            xtm = torch.empty((t_len, b, n, fin), dtype=torch.float32, device=x.device)
            x = xtm
        else:
            x = x.contiguous()
            _check(lib, x, "inputs")
        if h0 is not None:
            h0 = h0.contiguous()
            _check(lib, h0, "initial_hidden_state")
        need_grad = track and any(ctx.needs_input_grad)      # track: grad mode of the caller (off inside forward)
        pack = pack_cell(wg, bg, wc, bc, fin, h, m)
        ctx.params = (wg, bg, wc, bc)
        dev = x.device
        s = t_len * b
        if ready is not None:
            planes = ready["hpl"]                               # (M-1, T+1, B, N, Fin): slots 1..T are P_m x
            planes_ptr = planes.data_ptr() + 4 * b * n * fin
        else:
            planes = torch.empty((m - 1, s, n, fin), dtype=torch.float32, device=dev)
            planes_ptr = planes.data_ptr()
        hext = torch.empty((t_len + 1, b, n * h), dtype=torch.float32, device=dev)
        if need_grad:
            rs, us, cs, rhs = (torch.empty((t_len, b, n * h), dtype=torch.float32, device=dev) for _ in range(4))
            hpl, rhpl = (torch.empty((m - 1, t_len + 1, b, n, h), dtype=torch.float32, device=dev) for _ in range(2))
        else:
            rs = us = cs = rhs = hpl = rhpl = None
        ws = torch.empty(lib.query("eeg_dcrnn_layer_fwd_ws_floats", ctypes.byref(dims)), dtype=torch.float32, device=dev)
        lib.call("eeg_dcrnn_layer_fwd", ctypes.byref(dims), _p(xsrc if xsrc is not None else x), _p(xtm), _p(h0), _p(p),
                 _p(pack), planes_ptr, _p(hext), _p(rs), _p(us), _p(cs), _p(rhs), _p(hpl), _p(rhpl), _p(ws), _stream(x))
        hseq = hext[1:]
        if hpl is not None:
            _hop_planes[hseq.data_ptr()] = {"key": (t_len, b, n, h, m, p.data_ptr(), p_batched), "hpl": hpl,
                                            "hext": hext, "version": hext._version}
        if lengths is not None:
            lengths = lengths.to(device=dev, dtype=torch.int64).contiguous()
            hsel = torch.empty((b, n * h), dtype=torch.float32, device=dev)
            lib.call("eeg_dcrnn_gather_last", _p(hseq), _p(lengths), t_len, b, n * h, _p(hsel), _stream(x))
        else:
            hsel = hseq[t_len - 1].clone()
        if need_grad:
            ctx.save_for_backward(x, p, pack, planes, hext, rs, us, cs, rhs, hpl, rhpl, lengths)
            ctx.meta = (t_len, b, n, h, fin, m, act, p_batched, h0 is not None, ready is not None)
            ctx.set_materialize_grads(False)
        return hseq, hsel

    @staticmethod
    def backward(ctx, d_hseq, d_hsel):
        lib = _lib.get_lib()
        x, p, pack, planes, hext, rs, us, cs, rhs, hpl, rhpl, lengths = ctx.saved_tensors
        t_len, b, n, h, fin, m, act, p_batched, has_h0, planes_ready = ctx.meta
        dims = LayerDims(t_len, b, n, h, fin, m, act, p_batched)
        planes_ptr = planes.data_ptr()
        if planes_ready:                                        # the layer below's Hplanes, slots 1..T
            dims.x_planes_ready = 1
            dims.x_plane_stride = (t_len + 1) * b * n * fin
            planes_ptr += 4 * b * n * fin
        dev = x.device
        need_dx = ctx.needs_input_grad[0]
        need_dh0 = has_h0 and ctx.needs_input_grad[1]
        if d_hseq is not None:
            d_hseq = d_hseq.contiguous()
        if d_hsel is not None:
            d_hsel = d_hsel.contiguous()
        d_at_end = d_hsel if lengths is None else None
        d_at_len = d_hsel if lengths is not None else None
        dx = torch.empty_like(x) if need_dx else None
        dh0 = torch.empty((b, n * h), dtype=torch.float32, device=dev) if need_dh0 else None
        rows = (fin + h) * m
        shapes = ((rows, 2 * h), (2 * h,), (rows, h), (h,))
        sunk = [GradSink.take(q) for q in ctx.params]           # written in place -> nothing for autograd to add
        dwg, dbg, dwc, dbc = (t if t is not None else torch.empty(sh, dtype=torch.float32, device=dev)
                              for t, sh in zip(sunk, shapes))
        ws = torch.empty(lib.query("eeg_dcrnn_layer_bwd_ws_floats", ctypes.byref(dims), 1 if need_dx else 0),
                         dtype=torch.float32, device=dev)
        lib.call("eeg_dcrnn_layer_bwd", ctypes.byref(dims), _p(x), _p(p), _p(pack), planes_ptr, _p(hext), _p(rs),
                 _p(us), _p(cs), _p(rhs), _p(hpl), _p(rhpl), _p(d_hseq), _p(d_at_end), _p(d_at_len), _p(lengths), _p(dx), _p(dh0),
                 _p(dwg), _p(dbg), _p(dwc), _p(dbc), _p(ws), _stream(x))
        ret = [None if t is not None else g for t, g in zip(sunk, (dwg, dbg, dwc, dbc))]
        return (dx, dh0, None, *ret, None, None, None, None, None, None, None)


def dcgru_layer(x, h0, p, p_batched, wg, bg, wc, bc, n, h, m, activation="tanh", lengths=None):
    """Run one DCGRU layer over x (T,B,N,Fin).  Returns (hseq (T,B,N*H), hsel (B,N*H))."""
    act = ACT_CODES.get(activation, 1)
    lib = _lib.get_lib()
    if not lib.query("eeg_dcrnn_supported", n, h, x.shape[3], m):
        raise RuntimeError("eeg_gnn_ssl_amd: " + lib.last_error())
    return _DCGRULayerFn.apply(x, h0, p, wg, bg, wc, bc, lengths, p_batched, n, h, m, act, torch.is_grad_enabled())


class _DecoderFn(torch.autograd.Function):
    """DCGRUDecoder.forward (model.py:160-204) as one operator: T autoregressive steps through L cells
    and the projection; explicit BPTT backward with all parameter gradients hoisted over the T steps.

    inputs : targets (T,B,N*Dout) or None, h0 (L,B,N*H), P, teacher (tuple of T bools or None),
             first cell (wg,bg,wc,bc), shared cell (wg,bg,wc,bc) or Nones when L == 1, Wp (Dout,H), bp (Dout)
    output : (T,B,N*Dout)"""

    @staticmethod
    def forward(ctx, targets, h0, p, wg0, bg0, wc0, bc0, wg1, bg1, wc1, bc1, wp, bp, teacher, meta):
        lib = _lib.get_lib()
        t_len, b, n, h, dout, m, n_layers, act, p_batched = meta
        dev = h0.device
        h0 = h0.contiguous()
        wp, bp = wp.detach().contiguous(), bp.detach().contiguous()
        for t, nm in ((h0, "initial_hidden_state"), (p, "P"), (wp, "projection_layer.weight"), (bp, "projection_layer.bias")):
            _check(lib, t, nm)
        if tuple(wp.shape) != (dout, h) or tuple(bp.shape) != (dout,):
            raise RuntimeError(f"projection_layer shapes {tuple(wp.shape)}, {tuple(bp.shape)} do not match ({dout}, {h})")
        use_tf = teacher is not None and any(teacher)
        if use_tf:
            targets = targets.contiguous()
            _check(lib, targets, "inputs (teacher-forcing targets)")
        packs = [pack_cell(wg0, bg0, wc0, bc0, dout, h, m)]
        if n_layers > 1:
            packs += [pack_cell(wg1, bg1, wc1, bc1, h, h, m)] * (n_layers - 1)
        dims = DecoderDims(t_len, b, n, h, dout, m, n_layers, act, p_batched)
        out = torch.empty((t_len, b, n * dout), dtype=torch.float32, device=dev)
        saved = torch.empty(lib.query("eeg_dcrnn_decoder_saved_floats", ctypes.byref(dims)), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.query("eeg_dcrnn_decoder_fwd_ws_floats", ctypes.byref(dims)), dtype=torch.float32, device=dev)
        tf_arr = (ctypes.c_int32 * t_len)(*[1 if (use_tf and teacher[i]) else 0 for i in range(t_len)]) if use_tf else None
        pk_arr = (ctypes.c_void_p * n_layers)(*[q.data_ptr() for q in packs])
        lib.call("eeg_dcrnn_decoder_fwd", ctypes.byref(dims), _p(targets) if use_tf else None, tf_arr, _p(h0), _p(p), pk_arr,
                 _p(wp), _p(bp), _p(out), _p(saved), _p(ws), _stream(h0))
        ctx.save_for_backward(p, saved, wp, *packs[:2])
        ctx.params = (wg0, bg0, wc0, bc0, wg1, bg1, wc1, bc1, wp, bp)
        ctx.meta, ctx.teacher = meta, (tuple(bool(v) for v in teacher) if use_tf else None)
        ctx.shapes = (wg0.shape, bg0.shape, wc0.shape, bc0.shape,
                      None if wg1 is None else (wg1.shape, bg1.shape, wc1.shape, bc1.shape))
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.get_lib()
        p, saved, wp, *packs = ctx.saved_tensors
        t_len, b, n, h, dout, m, n_layers, act, p_batched = ctx.meta
        dev = saved.device
        d_out = d_out.contiguous()
        dims = DecoderDims(t_len, b, n, h, dout, m, n_layers, act, p_batched)
        packs = [packs[0]] + ([packs[1]] * (n_layers - 1) if n_layers > 1 else [])
        new = lambda shape: torch.empty(shape, dtype=torch.float32, device=dev)   # noqa: E731
        s0 = ctx.shapes
        shapes = list(s0[:4]) + (list(s0[4]) if n_layers > 1 else [None] * 4) + [(dout, h), (dout,)]
        sunk = [GradSink.take(q) if (q is not None and sh is not None) else None for q, sh in zip(ctx.params, shapes)]
        bufs = [t if t is not None else (new(sh) if sh is not None else None) for t, sh in zip(sunk, shapes)]
        g0, g1, dwp, dbp = bufs[0:4], bufs[4:8], bufs[8], bufs[9]
        dh0 = new((n_layers, b, n * h))
        ws = new((lib.query("eeg_dcrnn_decoder_bwd_ws_floats", ctypes.byref(dims)),))
        teacher = ctx.teacher
        tf_arr = (ctypes.c_int32 * t_len)(*[1 if teacher[i] else 0 for i in range(t_len)]) if teacher else None
        arr = lambda k: (ctypes.c_void_p * n_layers)(*[(g0 if l == 0 else g1)[k].data_ptr() for l in range(n_layers)])  # noqa: E731
        pk_arr = (ctypes.c_void_p * n_layers)(*[q.data_ptr() for q in packs])
        lib.call("eeg_dcrnn_decoder_bwd", ctypes.byref(dims), tf_arr, _p(p), pk_arr, _p(wp), _p(saved), _p(d_out), _p(dh0),
                 arr(0), arr(1), arr(2), arr(3), _p(dwp), _p(dbp), _p(ws), _stream(saved))
        ret = [None if t is not None else g for t, g in zip(sunk, bufs)]        # sunk: already in the caller's buffer
        return (None, dh0, None, *ret, None, None)


def dcgru_decoder(targets, h0, p, p_batched, first_cell, shared_cell, wp, bp, n, h, dout, m, n_layers,
                  activation="tanh", teacher=None):
    """Run the whole decoder: returns (T,B,N*Dout).  first_cell / shared_cell = (wg, bg, wc, bc)."""
    lib = _lib.get_lib()
    act = ACT_CODES.get(activation, 1)
    for fin in (dout, h):
        if not lib.query("eeg_dcrnn_supported", n, h, fin, m):
            raise RuntimeError("eeg_gnn_ssl_amd: " + lib.last_error())
    t_len, b = targets.shape[0], targets.shape[1]
    meta = (t_len, b, n, h, dout, m, n_layers, act, p_batched)
    sc = shared_cell if shared_cell is not None else (None, None, None, None)
    return _DecoderFn.apply(targets, h0, p, *first_cell, *sc, wp, bp, teacher, meta)


class _ClsHeadFn(torch.autograd.Function):
    """model.py:267-270 after dropout: per-node Linear(H->C) on relu(z), max over nodes."""

    @staticmethod
    def forward(ctx, z, w, bias):
        lib = _lib.get_lib()
        z = z.contiguous()
        w = w.contiguous()
        bias = bias.contiguous()
        for t, nm in ((z, "last_out"), (w, "fc.weight"), (bias, "fc.bias")):
            _check(lib, t, nm)
        b, n, h = z.shape
        c = w.shape[0]
        logits = torch.empty((b, c), dtype=torch.float32, device=z.device)
        arg = torch.empty((b, c), dtype=torch.int32, device=z.device)
        lib.call("eeg_dcrnn_cls_head_fwd", _p(z), _p(w), _p(bias), b, n, h, c, _p(logits), _p(arg), _stream(z))
        ctx.save_for_backward(z, w, arg)
        ctx.params = (w, bias)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.get_lib()
        z, w, arg = ctx.saved_tensors
        b, n, h = z.shape
        c = w.shape[0]
        dlogits = dlogits.contiguous()
        dz = torch.empty_like(z)
        sunk = [GradSink.take(q) for q in ctx.params]
        dw = sunk[0] if sunk[0] is not None else torch.empty_like(w)
        db = sunk[1] if sunk[1] is not None else torch.empty((c,), dtype=torch.float32, device=z.device)
        lib.call("eeg_dcrnn_cls_head_bwd", _p(z), _p(w), _p(dlogits), _p(arg), b, n, h, c, _p(dz), _p(dw), _p(db), _stream(z))
        return dz, (None if sunk[0] is not None else dw), (None if sunk[1] is not None else db)


def cls_head(z, w, bias):
    return _ClsHeadFn.apply(z, w, bias)


def gather_last(htop: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """utils.last_relevant_pytorch on a time-major (T,B,D) tensor, no host sync (forward only)."""
    lib = _lib.get_lib()
    htop = htop.contiguous()
    _check(lib, htop, "output")
    t_len, b, d = htop.shape
    lengths = lengths.to(device=htop.device, dtype=torch.int64).contiguous()
    out = torch.empty((b, d), dtype=torch.float32, device=htop.device)
    lib.call("eeg_dcrnn_gather_last", _p(htop), _p(lengths), t_len, b, d, _p(out), _stream(htop))
    return out


class _BCELogitsFn(torch.autograd.Function):
    """nn.BCEWithLogitsLoss() (mean): value and dlogits from one HIP launch (train.py:203-204)."""

    @staticmethod
    def forward(ctx, logits, y):
        lib = _lib.get_lib()
        x = logits.contiguous().view(-1)
        yy = y.to(torch.float32).contiguous().view(-1)
        _check(lib, x, "logits")
        _check(lib, yy, "targets")
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        lib.call("eeg_dcrnn_bce_logits", _p(x), _p(yy), x.numel(), _p(loss), _p(dx), _stream(x))
        ctx.save_for_backward(dx)
        ctx.shape = logits.shape
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        (dx,) = ctx.saved_tensors
        return (dx * dloss).view(ctx.shape), None


class _CELogitsFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean) on (B,C) logits and int64 class targets (train.py:205-206)."""

    @staticmethod
    def forward(ctx, logits, y):
        lib = _lib.get_lib()
        x = logits.contiguous()
        yy = y.to(torch.int64).contiguous()
        _check(lib, x, "logits")
        _check(lib, yy, "targets", torch.int64)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        lib.call("eeg_dcrnn_ce_logits", _p(x), _p(yy), x.shape[0], x.shape[1], _p(loss), _p(dx), _stream(x))
        ctx.save_for_backward(dx)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        (dx,) = ctx.saved_tensors
        return dx * dloss, None


class _MaskedLossFn(torch.autograd.Function):
    """utils.compute_regression_loss (utils.py:431-495): masked MAE / masked RMSE with an optional
    scalar StandardScaler inverse transform; value and gradient from three small HIP launches."""

    @staticmethod
    def forward(ctx, pred, y, mean, std, mask_val, kind):
        lib = _lib.get_lib()
        p = pred.contiguous()
        t = y.to(torch.float32).contiguous()
        _check(lib, p, "y_predicted")
        _check(lib, t, "y_true")
        if p.shape != t.shape:
            raise RuntimeError(f"y_predicted {tuple(p.shape)} and y_true {tuple(t.shape)} differ in shape")
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        dp = torch.empty_like(p)
        ws = torch.empty(lib.query("eeg_dcrnn_masked_loss_ws_floats"), dtype=torch.float32, device=p.device)
        scaled = mean is not None
        lib.call("eeg_dcrnn_masked_loss", _p(p), _p(t), p.numel(), 1 if scaled else 0, float(mean) if scaled else 0.0,
                 float(std) if scaled else 1.0, float(mask_val), int(kind), _p(loss), _p(dp), _p(ws), _stream(p))
        ctx.save_for_backward(dp)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        (dp,) = ctx.saved_tensors
        return dp * dloss, None, None, None, None, None


def masked_regression_loss(y_predicted, y_true, mean=None, std=None, loss_fn="mae", mask_val=0.0):
    """Only the exact string 'mae' selects the MAE (utils.py:489-495); anything else is the masked RMSE."""
    return _MaskedLossFn.apply(y_predicted, y_true, mean, std, mask_val, 0 if loss_fn == "mae" else 1)


def bce_with_logits(logits, y):
    return _BCELogitsFn.apply(logits, y)


def cross_entropy(logits, y):
    return _CELogitsFn.apply(logits, y)


def clip_adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay, max_norm, grad_scale, ws,
                   norm_out=None):
    """Fused clip_grad_norm_ + Adam (coupled L2) over flat fp32 buffers (train.py:273-275)."""
    lib = _lib.get_lib()
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _check(lib, t, nm)
    lib.call("eeg_dcrnn_clip_adam", _p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), params.numel(), float(max_norm),
             float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), float(grad_scale),
             _p(ws), _p(norm_out), _stream(params))
