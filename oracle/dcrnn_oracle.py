"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (pure torch eager, functional, op-for-op) of the DCRNN hot path of
tsy935/eeg-gnn-ssl.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import this file, and only as the checker / CPU baseline.

Parity status: PINNED.  `tests/test_oracle_vs_golden.py` checks every function here
against vectors produced by importing the genuine reference modules in the build
container (`tests/golden/make_golden.py`, outputs committed under `tests/golden/`).
The reference itself ships no tests and no golden vectors.

Each function cites the reference file:line it restates.  The quirks it reproduces
on purpose are SURVEY.md §7 Q1-Q9 (carried x0 across supports, f-major/hop-minor
weight rows, ignored bias_start, shared decoder cell, RMSE-for-"MAE", ...).

Parameters travel as a flat dict keyed by the reference's `state_dict` names, e.g.
`encoder.encoding_cells.0.dconv_gate.weight`, so reference checkpoints load as-is.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration (defaults = reference args.py:74-119)
# --------------------------------------------------------------------------------------
@dataclass
class DCRNNConfig:
    num_nodes: int = 19
    num_rnn_layers: int = 2
    rnn_units: int = 64
    input_dim: int = 100
    output_dim: int = 100
    max_diffusion_step: int = 2
    dcgru_activation: str = "tanh"
    filter_type: str = "laplacian"
    dropout: float = 0.0
    num_classes: int = 1
    cl_decay_steps: int = 3000
    use_curriculum_learning: bool = False

    @property
    def num_supports(self) -> int:  # cell.py:151-158
        return 2 if self.filter_type == "dual_random_walk" else 1

    @property
    def num_matrices(self) -> int:  # cell.py:35
        return self.num_supports * self.max_diffusion_step + 1


# --------------------------------------------------------------------------------------
# graph supports (float64 numpy, like the reference's scipy path)
# --------------------------------------------------------------------------------------
def scaled_laplacian(adj: np.ndarray, lambda_max: Optional[float] = None) -> np.ndarray:
    """utils.py:205-217 + 240-255 (calculate_scaled_laplacian, undirected=True).

    L~ = 2 L / lambda_max - I,  L = I - D^-1/2 A D^-1/2, A symmetrised by max.
    `lambda_max=None` -> largest-magnitude eigenvalue (reference: scipy eigsh 'LM').
    Returned dense float64 (the reference converts to FloatTensor afterwards)."""
    a = np.maximum(adj, adj.T)          # dtype preserved: the reference's float32 asset keeps the
    d = a.sum(axis=1)                   # degree / D^-1/2 A D^-1/2 arithmetic in float32
    with np.errstate(divide="ignore"):
        dis = np.power(d, -0.5)
    dis[np.isinf(dis)] = 0.0
    # reference: I - (A D^-1/2)^T D^-1/2  ==  I - D^-1/2 A^T D^-1/2   (float64 from here on)
    lap = np.eye(a.shape[0]) - (a * dis[None, :]).T * dis[None, :]
    if lambda_max is None:
        ev = np.linalg.eigvalsh((lap + lap.T) * 0.5)
        lambda_max = ev[np.argmax(np.abs(ev))]
    return (2.0 / lambda_max) * lap - np.eye(a.shape[0])


def random_walk(adj: np.ndarray) -> np.ndarray:
    """utils.py:220-230 (calculate_random_walk_matrix): D^-1 A, rows with zero degree -> 0."""
    a = np.asarray(adj)                 # dtype preserved (float32 for the per-clip graphs)
    d = a.sum(axis=1)
    with np.errstate(divide="ignore"):
        dinv = np.power(d, -1.0)
    dinv[np.isinf(dinv)] = 0.0
    return dinv[:, None] * a


def compute_supports(adj: np.ndarray, filter_type: str) -> List[Tensor]:
    """dataloader_detection.py:335-354 (_compute_supports) -> list of float32 (N,N) tensors."""
    if filter_type == "laplacian":
        mats = [scaled_laplacian(adj, lambda_max=None)]
    elif filter_type == "random_walk":
        mats = [random_walk(adj).T]
    elif filter_type == "dual_random_walk":
        mats = [random_walk(adj).T, random_walk(adj.T).T]
    else:  # reference falls back to lambda_max=2
        mats = [scaled_laplacian(adj, lambda_max=2)]
    return [torch.from_numpy(np.ascontiguousarray(m)).to(torch.float32) for m in mats]


def keep_topk(adj: np.ndarray, top_k: int = 3, directed: bool = True) -> np.ndarray:
    """data_utils.py:174-200."""
    work = adj.copy()
    np.fill_diagonal(work, 0)
    idx = (-work).argsort(axis=-1)[:, :top_k]
    mask = np.eye(adj.shape[0], dtype=bool)
    for i in range(idx.shape[0]):
        for j in idx[i]:
            mask[i, j] = True
            if not directed:
                mask[j, i] = True
    return mask * adj


def correlation_adjacency(clip: np.ndarray, top_k: int = 3) -> np.ndarray:
    """dataloader_detection.py:258-307 (_get_indiv_graphs) without node swaps.

    clip: (T, N, D).  |normalised 'valid' cross-correlation| of equal-length signals is
    one dot product, i.e. the |cosine Gram| of the (N, T*D) matrix (data_utils.py:203-222);
    diagonal 1; then top-k directed sparsification.  float32 result like the reference."""
    n = clip.shape[1]
    flat = np.transpose(clip, (1, 0, 2)).reshape(n, -1)
    adj = np.eye(n, dtype=np.float32)
    for i in range(n):
        for j in range(i + 1, n):
            xc = np.correlate(flat[i], flat[j], mode="valid")  # == scipy.signal.correlate 'valid'
            cxx = np.sum(np.abs(flat[i]) ** 2)
            cyy = np.sum(np.abs(flat[j]) ** 2)
            if cxx != 0 and cyy != 0:
                xc = xc / (cxx * cyy) ** 0.5
            adj[i, j] = xc[0]
            adj[j, i] = xc[0]
    adj = np.abs(adj)
    return keep_topk(adj, top_k=top_k, directed=True)


def fft_features(raw: np.ndarray, window: int = 200) -> np.ndarray:
    """data_utils.py:13-35 (computeFFT) applied per window as dataloader_detection.py:57-71 does.

    raw (N, T*window) -> (T, N, window//2) float64 log amplitudes of the positive-frequency half of the
    FFT of every `window`-sample step; exact-zero amplitudes are replaced by 1e-8 before the log."""
    n_ch, total = raw.shape
    steps = total // window
    out = np.empty((steps, n_ch, window // 2), dtype=np.float64)
    for t in range(steps):
        spec = np.fft.fft(np.asarray(raw[:, t * window:(t + 1) * window], dtype=np.float64), n=window, axis=-1)
        amp = np.abs(spec[:, :window // 2])
        amp[amp == 0.0] = 1e-8
        out[t] = np.log(amp)
    return out


# --------------------------------------------------------------------------------------
# diffusion graph convolution and DCGRU cell
# --------------------------------------------------------------------------------------
def hop_stack(supports: Sequence[Tensor], x0: Tensor, k_max: int) -> Tensor:
    """cell.py:76-93.  x0: (B,N,F); supports: each (N,N) or (B,N,N).  Returns (B,M,N,F).

    Q1: the reference's `x1, x0 = x2, x1` leaves x0 modified when it moves on to the
    next support, so with two supports the 4th/5th hops start from S1*X, not X."""
    hops = [x0]
    if k_max > 0:
        prev = x0
        for sup in supports:
            cur = torch.matmul(sup, prev)
            hops.append(cur)
            for _ in range(2, k_max + 1):
                nxt = 2 * torch.matmul(sup, cur) - prev
                hops.append(nxt)
                cur, prev = nxt, cur
    return torch.stack(hops, dim=1)


def diffusion_conv(supports: Sequence[Tensor], inputs: Tensor, state: Tensor,
                   weight: Tensor, biases: Tensor, num_nodes: int, k_max: int) -> Tensor:
    """cell.py:66-118 (DiffusionGraphConv.forward).  inputs (B,N*Din), state (B,N*H) ->
    (B, N*O).  Q2: weight rows are ordered f*M + m (feature-major, hop-minor)."""
    b = inputs.shape[0]
    xs = torch.cat([inputs.reshape(b, num_nodes, -1), state.reshape(b, num_nodes, -1)], dim=2)
    hops = hop_stack(supports, xs, k_max)                     # (B,M,N,F)
    m, f = hops.shape[1], hops.shape[3]
    flat = hops.permute(0, 2, 3, 1).reshape(b * num_nodes, f * m)   # cell.py:98-114
    out = torch.matmul(flat, weight) + biases                 # cell.py:116-117
    return out.reshape(b, num_nodes * weight.shape[1])


def dcgru_cell(supports: Sequence[Tensor], inputs: Tensor, state: Tensor,
               wg: Tensor, bg: Tensor, wc: Tensor, bc: Tensor,
               num_nodes: int, num_units: int, k_max: int, activation: str = "tanh") -> Tensor:
    """cell.py:182-210 (DCGRUCell.forward).  Returns the new state (== output).

    Q3: first H gate columns are r, last H are u; candidate sees r*h; h' = u*h + (1-u)*c.
    Q4: the `bias_start=1.0` the reference passes here is ignored by DiffusionGraphConv."""
    b = inputs.shape[0]
    g = torch.sigmoid(diffusion_conv(supports, inputs, state, wg, bg, num_nodes, k_max))
    g = g.reshape(b, num_nodes, 2 * num_units)
    r = g[..., :num_units].reshape(b, num_nodes * num_units)
    u = g[..., num_units:].reshape(b, num_nodes * num_units)
    c = diffusion_conv(supports, inputs, r * state, wc, bc, num_nodes, k_max)
    c = torch.tanh(c) if activation == "tanh" else torch.relu(c)
    return u * state + (1 - u) * c


def _cell_params(params: Dict[str, Tensor], prefix: str):
    return (params[prefix + ".dconv_gate.weight"], params[prefix + ".dconv_gate.biases"],
            params[prefix + ".dconv_candidate.weight"], params[prefix + ".dconv_candidate.biases"])


# --------------------------------------------------------------------------------------
# sequence loops
# --------------------------------------------------------------------------------------
def encoder_forward(params: Dict[str, Tensor], cfg: DCRNNConfig, inputs: Tensor,
                    init_hidden: Tensor, supports: Sequence[Tensor],
                    prefix: str = "encoder") -> Tuple[Tensor, Tensor]:
    """model.py:81-102 (DCRNNEncoder.forward): layer-major, then time.  inputs (T,B,N,Din),
    init_hidden (L,B,N*H) -> (final hidden per layer (L,B,N*H), top-layer sequence (T,B,N*H))."""
    t_len, b = inputs.shape[0], inputs.shape[1]
    cur = inputs.reshape(t_len, b, -1)
    finals = []
    for layer in range(cfg.num_rnn_layers):
        wg, bg, wc, bc = _cell_params(params, f"{prefix}.encoding_cells.{layer}")
        h = init_hidden[layer]
        outs = []
        for t in range(t_len):
            h = dcgru_cell(supports, cur[t], h, wg, bg, wc, bc, cfg.num_nodes, cfg.rnn_units,
                           cfg.max_diffusion_step, cfg.dcgru_activation)
            outs.append(h)
        finals.append(h)
        cur = torch.stack(outs, dim=0)
    return torch.stack(finals, dim=0), cur


def decoder_forward(params: Dict[str, Tensor], cfg: DCRNNConfig, targets: Tensor,
                    init_hidden: Tensor, supports: Sequence[Tensor],
                    teacher_force_mask: Optional[Sequence[bool]] = None,
                    prefix: str = "decoder", dropout_masks: Optional[Tensor] = None) -> Tensor:
    """model.py:149-204 (DCGRUDecoder.forward): time-major autoregressive loop, GO = zeros,
    per-step Linear(H->Dout) per node, output fed back.  targets (T,B,N,Dout) are only used
    when teacher forcing; the reference draws `random.random() < ratio` per step — the draw
    is passed in here as an explicit per-step mask so the oracle stays deterministic.
    Q6: decoding_cells.l for every l >= 1 carry the same tensors in a reference state_dict
    (one shared cell object); the dict-of-names interface reproduces that automatically.
    Dropout (model.py:191-192, `self.projection_layer(self.dropout(output...))`): nn.Dropout in training mode multiplies by a
    Bernoulli keep-mask scaled by 1/(1-p), a fresh one every step.  The draw is an input here like the teacher-forcing flags:
    dropout_masks (T,B,N,H) holds the mask x scale factors (None = eval mode / p = 0)."""
    t_len, b = targets.shape[0], targets.shape[1]
    tgt = targets.reshape(t_len, b, -1)
    w_proj = params[f"{prefix}.projection_layer.weight"]
    b_proj = params[f"{prefix}.projection_layer.bias"]
    hidden = [init_hidden[l] for l in range(cfg.num_rnn_layers)]
    cur_in = torch.zeros(b, cfg.num_nodes * cfg.output_dim, dtype=targets.dtype, device=targets.device)
    outs = []
    for t in range(t_len):
        x = cur_in
        for layer in range(cfg.num_rnn_layers):
            wg, bg, wc, bc = _cell_params(params, f"{prefix}.decoding_cells.{layer}")
            hidden[layer] = dcgru_cell(supports, x, hidden[layer], wg, bg, wc, bc, cfg.num_nodes,
                                       cfg.rnn_units, cfg.max_diffusion_step, cfg.dcgru_activation)
            x = hidden[layer]
        top = x.reshape(b, cfg.num_nodes, cfg.rnn_units)
        if dropout_masks is not None:
            top = top * dropout_masks[t].reshape(b, cfg.num_nodes, cfg.rnn_units)
        proj = torch.matmul(top, w_proj.t()) + b_proj
        proj = proj.reshape(b, cfg.num_nodes * cfg.output_dim)
        outs.append(proj)
        if teacher_force_mask is not None and teacher_force_mask[t]:
            cur_in = tgt[t]
        else:
            cur_in = proj
    return torch.stack(outs, dim=0)


def last_relevant(output: Tensor, lengths: Tensor) -> Tensor:
    """utils.py:346-357 (batch_first=True): output (B,T,D) gathered at t = len-1 -> (B,D)."""
    idx = (lengths.to(device=output.device, dtype=torch.int64) - 1).view(-1, 1, 1).expand(-1, 1, output.shape[2])
    return output.gather(1, idx).squeeze(1)


def classification_forward(params: Dict[str, Tensor], cfg: DCRNNConfig, input_seq: Tensor,
                           seq_lengths: Tensor, supports: Sequence[Tensor],
                           dropout_mask: Optional[Tensor] = None) -> Tensor:
    """model.py:235-272 (DCRNNModel_classification.forward).  input_seq (B,T,N,Din) -> (B,C).
    Q7: fc(relu(dropout(h_last))) per node, then max over nodes.  dropout_mask (B,N,H): the training-mode nn.Dropout draw of
    model.py:267 as mask x 1/(1-p) factors (README.md:83 trains the 4-class model with --dropout 0.5); None = eval / p = 0."""
    b = input_seq.shape[0]
    x = input_seq.transpose(0, 1)
    h0 = torch.zeros(cfg.num_rnn_layers, b, cfg.num_nodes * cfg.rnn_units, dtype=input_seq.dtype, device=input_seq.device)
    _, top = encoder_forward(params, cfg, x, h0, supports)
    last = last_relevant(top.transpose(0, 1), seq_lengths).view(b, cfg.num_nodes, cfg.rnn_units)
    if dropout_mask is not None:
        last = last * dropout_mask.reshape(b, cfg.num_nodes, cfg.rnn_units)
    logits = torch.matmul(torch.relu(last), params["fc.weight"].t()) + params["fc.bias"]
    return logits.max(dim=1).values


def next_time_pred_forward(params: Dict[str, Tensor], cfg: DCRNNConfig, encoder_inputs: Tensor,
                           decoder_inputs: Tensor, supports: Sequence[Tensor],
                           teacher_force_mask: Optional[Sequence[bool]] = None,
                           dropout_masks: Optional[Tensor] = None) -> Tensor:
    """model.py:313-360 (DCRNNModel_nextTimePred.forward) -> (B,T_out,N,Dout).  dropout_masks: see decoder_forward."""
    b, t_out, n, _ = decoder_inputs.shape
    enc_in = encoder_inputs.transpose(0, 1)
    dec_in = decoder_inputs.transpose(0, 1)
    h0 = torch.zeros(cfg.num_rnn_layers, b, cfg.num_nodes * cfg.rnn_units, dtype=encoder_inputs.dtype, device=encoder_inputs.device)
    enc_final, _ = encoder_forward(params, cfg, enc_in, h0, supports)
    out = decoder_forward(params, cfg, dec_in, enc_final, supports, teacher_force_mask, dropout_masks=dropout_masks)
    return out.reshape(t_out, b, n, -1).transpose(0, 1)


def compute_sampling_threshold(cl_decay_steps: float, global_step: float) -> float:
    """utils.py:385-390."""
    return cl_decay_steps / (cl_decay_steps + math.exp(global_step / cl_decay_steps))


# --------------------------------------------------------------------------------------
# losses that seed backward
# --------------------------------------------------------------------------------------
def bce_with_logits(logits: Tensor, y: Tensor) -> Tensor:
    """train.py:203-204,266-267: nn.BCEWithLogitsLoss()(logits.view(-1), y), mean."""
    return torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), y)


def cross_entropy(logits: Tensor, y: Tensor) -> Tensor:
    """train.py:205-206,268: nn.CrossEntropyLoss()(logits, y), mean."""
    return torch.nn.functional.cross_entropy(logits, y)


def masked_mae(y_pred: Tensor, y_true: Tensor, mask_val: float = 0.0) -> Tensor:
    """utils.py:431-442."""
    w = (y_true != mask_val).to(y_pred.dtype)
    w = w / w.mean()
    loss = (y_pred - y_true).abs() * w
    loss = torch.where(loss != loss, torch.zeros_like(loss), loss)
    return loss.mean()


def masked_rmse(y_pred: Tensor, y_true: Tensor, mask_val: float = 0.0) -> Tensor:
    """utils.py:445-457 (`masked_mse_loss`, which really returns a masked RMSE)."""
    w = (y_true != mask_val).to(y_pred.dtype)
    w = w / w.mean()
    loss = (y_pred - y_true).pow(2) * w
    loss = torch.where(loss != loss, torch.zeros_like(loss), loss)
    return torch.sqrt(loss.mean())


def regression_loss(y_true: Tensor, y_pred: Tensor, mean: Optional[float] = None,
                    std: Optional[float] = None, loss_fn: str = "mae") -> Tensor:
    """utils.py:460-495 (compute_regression_loss) with a scalar StandardScaler
    (utils.py:393-428).  Q9: only the exact string 'mae' selects MAE; train_ssl.py:168
    passes "MAE" and therefore trains on the masked RMSE."""
    if mean is not None:
        y_true = y_true * std + mean
        y_pred = y_pred * std + mean
    if loss_fn == "mae":
        return masked_mae(y_pred, y_true)
    return masked_rmse(y_pred, y_true)


# --------------------------------------------------------------------------------------
# parameter construction (shapes/names = reference state_dict; init = cell.py:40-48)
# --------------------------------------------------------------------------------------
def _cell_shapes(cfg: DCRNNConfig, in_dim: int):
    m = cfg.num_matrices
    rows = (in_dim + cfg.rnn_units) * m
    return {
        "dconv_gate.weight": (rows, 2 * cfg.rnn_units), "dconv_gate.biases": (2 * cfg.rnn_units,),
        "dconv_candidate.weight": (rows, cfg.rnn_units), "dconv_candidate.biases": (cfg.rnn_units,),
    }


def param_shapes(cfg: DCRNNConfig, model: str) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape for `classification` (model.py:209-233) or `ssl`
    (model.py:278-311).  For ssl with L >= 3 the reference state_dict lists
    decoding_cells.1 .. L-1 separately although they are one shared cell (Q6)."""
    shapes: Dict[str, Tuple[int, ...]] = {}
    for layer in range(cfg.num_rnn_layers):
        in_dim = cfg.input_dim if layer == 0 else cfg.rnn_units
        for k, v in _cell_shapes(cfg, in_dim).items():
            shapes[f"encoder.encoding_cells.{layer}.{k}"] = v
    if model == "classification":
        shapes["fc.weight"] = (cfg.num_classes, cfg.rnn_units)
        shapes["fc.bias"] = (cfg.num_classes,)
    else:
        for layer in range(cfg.num_rnn_layers):
            in_dim = cfg.output_dim if layer == 0 else cfg.rnn_units
            for k, v in _cell_shapes(cfg, in_dim).items():
                shapes[f"decoder.decoding_cells.{layer}.{k}"] = v
        shapes["decoder.projection_layer.weight"] = (cfg.output_dim, cfg.rnn_units)
        shapes["decoder.projection_layer.bias"] = (cfg.output_dim,)
    return shapes


def init_params(cfg: DCRNNConfig, model: str, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Random parameters with the reference's init law (xavier-normal gain 1.414 for dconv
    weights, zero biases — cell.py:47-48; nn.Linear default for fc/projection)."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, Tensor] = {}
    shared: Dict[str, Tensor] = {}
    for name, shape in param_shapes(cfg, model).items():
        if ".decoding_cells." in name:
            layer = int(name.split(".")[2])
            if layer >= 2:  # Q6: same tensors as layer 1
                out[name] = shared[name.replace(f"decoding_cells.{layer}", "decoding_cells.1")]
                continue
        if name.endswith("dconv_gate.weight") or name.endswith("dconv_candidate.weight"):
            std = 1.414 * math.sqrt(2.0 / (shape[0] + shape[1]))
            t = torch.randn(shape, generator=g, dtype=torch.float32) * std
        elif name.endswith(".biases"):
            t = torch.zeros(shape)
        else:  # nn.Linear: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for both weight and bias
            fan_in = cfg.rnn_units
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        out[name] = t.to(dtype)
        shared[name] = out[name]
    return out
