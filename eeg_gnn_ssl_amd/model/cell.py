"""DCGRU cell modules with the reference's constructor/forward signatures and parameter names
(tsy935/eeg-gnn-ssl model/cell.py:17-225), backed by the MI355X HIP kernels.

`state_dict` keys and shapes are identical to the reference (`dconv_gate.weight`
((Fin+H)*M, 2H), `dconv_gate.biases` (2H,), `dconv_candidate.weight` ((Fin+H)*M, H),
`dconv_candidate.biases` (H,)), so reference checkpoints load unchanged.
"""
import math

import torch
import torch.nn as nn

from .. import ops


class DiffusionGraphConv(nn.Module):
    """One diffusion convolution (reference: cell.py:17-48).

    weight rows are ordered f*M + m (feature-major, hop-minor; cell.py:98-116).  Initialisation
    follows the reference: xavier-normal with gain 1.414 and constant `bias_start` biases.
    The arithmetic lives in the fused cell kernels (gate and candidate convolutions share the
    diffused input), which is why `DCGRUCell.forward` does not call this module's forward."""

    def __init__(self, num_supports, input_dim, hid_dim, num_nodes, max_diffusion_step, output_dim,
                 bias_start=0.0, filter_type="laplacian"):
        super().__init__()
        self._num_matrices = num_supports * max_diffusion_step + 1
        self._input_size = input_dim + hid_dim
        self._input_dim = input_dim
        self._hid_dim = hid_dim
        self._num_nodes = num_nodes
        self._max_diffusion_step = max_diffusion_step
        self._filter_type = filter_type
        self.weight = nn.Parameter(torch.empty(self._input_size * self._num_matrices, output_dim))
        self.biases = nn.Parameter(torch.empty(output_dim))
        nn.init.xavier_normal_(self.weight, gain=1.414)
        nn.init.constant_(self.biases, bias_start)

    def forward(self, supports, inputs, state, output_size, bias_start=0.0):
        """(B, N*Din), (B, N*H) -> (B, N*output_size); reference cell.py:66-118.

        HIP diffusion kernel + fp32-MFMA GEMM on the reference's weight layout (`torch.ops.eeg_dcrnn.dconv`);
        differentiable w.r.t. inputs, state, weight and biases like the reference module (the fused training
        path never calls it: gate and candidate convolutions share the diffused input there)."""
        b = inputs.shape[0]
        n, f = self._num_nodes, self._input_size
        if f % 4 != 0:
            raise RuntimeError(f"DiffusionGraphConv: input_dim + hid_dim = {f} must be a multiple of 4")
        x = torch.cat([inputs.reshape(b, n, -1), state.reshape(b, n, -1)], dim=2)
        p, p_batched = ops.hop_polys(supports, self._max_diffusion_step, b)
        out = ops.dconv(x, p, p_batched, self.weight, self.biases)
        return out.reshape(b, n * output_size)


class DCGRUCell(nn.Module):
    """Diffusion-convolutional GRU cell (reference: cell.py:121-225).

    forward(supports, inputs (B, N*Din), state (B, N*H)) -> (output, new_state), both (B, N*H):
        r, u = sigmoid(dconv_gate([x | h]));  c = act(dconv_candidate([x | r*h]));
        h' = u*h + (1-u)*c
    One call = one step of the persistent HIP sequence kernel (T = 1)."""

    def __init__(self, input_dim, num_units, max_diffusion_step, num_nodes, filter_type="laplacian",
                 nonlinearity="tanh", use_gc_for_ru=True):
        super().__init__()
        self._activation_name = "tanh" if nonlinearity == "tanh" else "relu"   # cell.py:146
        self._num_nodes = num_nodes
        self._num_units = num_units
        self._input_dim = input_dim
        self._max_diffusion_step = max_diffusion_step
        self._use_gc_for_ru = use_gc_for_ru
        if not use_gc_for_ru:
            raise NotImplementedError("use_gc_for_ru=False is a stub (`_fc` is `pass`) in the reference as well")
        if max_diffusion_step < 0:
            raise ValueError("max_diffusion_step must be >= 0")
        self._num_supports = 2 if filter_type == "dual_random_walk" else 1     # cell.py:151-158
        self._filter_type = filter_type
        common = dict(num_supports=self._num_supports, input_dim=input_dim, hid_dim=num_units,
                      num_nodes=num_nodes, max_diffusion_step=max_diffusion_step, filter_type=filter_type)
        self.dconv_gate = DiffusionGraphConv(output_dim=num_units * 2, **common)
        self.dconv_candidate = DiffusionGraphConv(output_dim=num_units, **common)

    @property
    def output_size(self):
        return self._num_nodes * self._num_units

    @property
    def num_matrices(self):
        return self._num_supports * self._max_diffusion_step + 1

    def _check_supports(self, supports):
        if len(supports) != self._num_supports:
            raise RuntimeError(f"filter_type={self._filter_type!r} expects {self._num_supports} support(s), "
                               f"got {len(supports)}")
        n = self._num_nodes
        for i, s in enumerate(supports):
            if s.dim() not in (2, 3) or s.shape[-1] != n or s.shape[-2] != n:
                raise RuntimeError(f"supports[{i}] has shape {tuple(s.shape)}, expected ({n}, {n}) or (B, {n}, {n})")

    def run_sequence(self, x, h0, p, p_batched, lengths=None, x_off=0, x_planes=None, want_hsel=True, basis=None, pack=None, spack=None):
        """x (T + x_off, B, N, Din) -> ops.LayerOut (hext (T+1,B,N*H), hsel (B,N*H), hpl); used by the encoder /
        decoder loops.  x_off = 1 with x_planes: x is the `hext` of the layer below and x_planes its `hpl`.
        basis: `ops.shared_spectral_basis` of the one symmetric support all clips share (or None).
        pack / spack: this cell's packs when the caller made them ahead (`ops.pack_cells`: all layers in one launch)."""
        return ops.dcgru_layer_ex(x, x_off, h0, p, p_batched, self.dconv_gate.weight, self.dconv_gate.biases,
                                  self.dconv_candidate.weight, self.dconv_candidate.biases,
                                  self._num_nodes, self._num_units, self.num_matrices,
                                  self._activation_name, lengths, x_planes, want_hsel, basis, pack, spack)

    def forward(self, supports, inputs, state):
        self._check_supports(supports)
        b = inputs.shape[0]
        p, p_batched = ops.hop_polys(supports, self._max_diffusion_step, b)
        x = inputs.reshape(1, b, self._num_nodes, self._input_dim)
        new_state = self.run_sequence(x, state, p, p_batched).hsel      # T = 1: h at the last (only) step
        return new_state, new_state

    def init_hidden(self, batch_size):
        return torch.zeros(batch_size, self._num_nodes * self._num_units,
                           device=self.dconv_gate.weight.device)
