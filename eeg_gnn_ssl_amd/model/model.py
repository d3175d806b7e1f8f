"""DCRNN encoder / decoder / task models with the reference's signatures, attribute names and
`state_dict` layout (tsy935/eeg-gnn-ssl model/model.py:48-360), backed by the MI355X kernels.

Differences that are deliberate and invisible to callers:
  * one persistent HIP launch per (layer, direction) replaces the Python `for t` loop of
    model.py:93-96 (the x-part of every diffusion convolution is hoisted out of the recurrence);
  * hop polynomials of the supports are built once per forward instead of per step;
  * the classification model gathers h at len-1 on the device (no `lengths.cpu()` sync,
    utils.py:347) and routes its gradient straight into the BPTT kernel.
"""
import random

import torch
import torch.nn as nn

from .. import ops
from .. import utils
from .cell import DCGRUCell


class _FusedDropout:
    """Mixin of the two modules that own an `nn.Dropout` in the reference (model.py:191,267).  The module keeps its
    `self.dropout = nn.Dropout(p)` attribute (same repr / `p` semantics), but the mask is generated inside the HIP kernel that
    consumes the dropped tensor from a device-resident Philox4x32-10 generator state (int64 {seed, offset}) that every
    dropping forward advances on the device -- a captured HIP graph therefore draws a fresh mask on every replay.

    The state is a NON-persistent buffer `_dropout_rng`, created in the constructor (so `.to(device)` moves it with the
    parameters and nothing is allocated or copied inside a forward that may be under graph capture).  Its seed derives from
    `torch.initial_seed()` -- `torch.manual_seed` before construction governs it -- without drawing from the global generator
    (later host-side draws, e.g. DataLoader shuffles, are where they would be in a reference run).  It is outside `state_dict`
    on purpose (reference checkpoints load with strict=True); a run that wants mask-reproducible resumption saves
    `dropout_rng_state()` next to the checkpoint and restores it with `set_dropout_seed(seed, offset)`."""

    def _init_dropout_rng(self, stream_id: int):
        self.register_buffer("_dropout_rng", ops.make_rng_state("cpu", stream_id), persistent=False)
        # a wrapper that broadcasts buffers (torch DistributedDataParallel, broadcast_buffers=True) must leave the generator state
        # alone: rank 0's state on every rank would mean identical masks / teacher-forcing flags everywhere, and a collective
        # inside a forward that may be under graph capture.  (The seed already mixes the rank in: ops.make_rng_state.)
        self._ddp_params_and_buffers_to_ignore = ["_dropout_rng"]

    def _drop_p(self) -> float:
        return float(self.dropout.p) if self.training else 0.0

    def set_dropout_seed(self, seed: int, offset: int = 0):
        """Re-seed the fused dropout generator (in place: captured graphs keep reading the same tensor)."""
        st = self._dropout_rng
        st.copy_(torch.tensor([int(seed), int(offset)], dtype=torch.int64), non_blocking=False)

    def dropout_rng_state(self):
        """(seed, offset) of the generator right now (synchronises with the device)."""
        seed, offset = self._dropout_rng.tolist()
        return int(seed), int(offset)

    def _rng_state(self, device) -> torch.Tensor:
        st = self._dropout_rng
        device = torch.device(device)
        if st.device.type != device.type or (device.index is not None and st.device.index != device.index):
            if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the fused dropout generator state is not on the device of the inputs: move the model "
                                   "with .to(device) before capturing its forward into a HIP graph")
            st = st.to(device)
            self._buffers["_dropout_rng"] = st
        return st


class DCRNNEncoder(nn.Module):
    """reference: model.py:48-109."""

    def __init__(self, input_dim, max_diffusion_step, hid_dim, num_nodes, num_rnn_layers,
                 dcgru_activation=None, filter_type="laplacian", device=None):
        super().__init__()
        self.hid_dim = hid_dim
        self.num_rnn_layers = num_rnn_layers
        self.num_nodes = num_nodes
        self.max_diffusion_step = max_diffusion_step
        self._device = device
        cells = []
        for layer in range(num_rnn_layers):
            cells.append(DCGRUCell(input_dim=input_dim if layer == 0 else hid_dim, num_units=hid_dim,
                                   max_diffusion_step=max_diffusion_step, num_nodes=num_nodes,
                                   nonlinearity=dcgru_activation, filter_type=filter_type))
        self.encoding_cells = nn.ModuleList(cells)

    def run(self, inputs, initial_hidden_state, supports, lengths=None, want_finals=True):
        """Layer-major pass (model.py:90-99).  Returns (finals (L,B,N*H) or None, top sequence (T,B,N*H),
        top state at t = lengths-1 (B,N*H) or None)."""
        t_len, b = inputs.shape[0], inputs.shape[1]
        self.encoding_cells[0]._check_supports(supports)
        p, p_batched = ops.hop_polys(supports, self.max_diffusion_step, b)
        # one symmetric support shared by all clips (given as an (N,N) tensor: the scaled Laplacian of the distance graph): the
        # hoisted x-part of every layer runs in its eigenbasis (K = Fin instead of M * Fin); else None = the general path
        basis = ops.shared_spectral_basis(supports, self.max_diffusion_step)
        cur, x_off, planes = inputs.reshape(t_len, b, self.num_nodes, -1), 0, None
        finals, top_sel, out = [], None, None
        # the weight packs of ALL layers in one launch (they change with every optimisation step); None: each layer packs its own
        packs = ops.pack_encoder_cells(self.encoding_cells, basis, self.num_nodes)
        for layer, cell in enumerate(self.encoding_cells):
            h0 = None if initial_hidden_state is None else initial_hidden_state[layer]
            is_top = layer == self.num_rnn_layers - 1
            # layers >= 1 read the `hext` of the layer below (slots 1..T) and take its hop planes as their own
            # (a layer's final state is only copied out when somebody reads it: `finals`, or the top state at len-1)
            out = cell.run_sequence(cur, h0, p, p_batched, lengths if is_top else None, x_off, planes,
                                    want_hsel=want_finals or (is_top and lengths is not None), basis=basis,
                                    pack=None if packs is None else packs[0][layer], spack=None if packs is None else packs[1][layer])
            if is_top and lengths is not None:
                top_sel = out.hsel
                finals.append(out.hext[t_len] if want_finals else None)
            else:
                finals.append(out.hsel if want_finals else None)
            cur, x_off, planes = out.hext.view(t_len + 1, b, self.num_nodes, self.hid_dim), 1, out.hpl
        return (torch.stack(finals, dim=0) if want_finals else None), out.hseq, top_sel

    def forward(self, inputs, initial_hidden_state, supports):
        """inputs (T,B,N,Din), initial_hidden_state (L,B,N*H) ->
        (output_hidden (L,B,N*H), current_inputs (T,B,N*H))"""
        finals, top, _ = self.run(inputs, initial_hidden_state, supports)
        return finals, top

    def init_hidden(self, batch_size):
        return torch.stack([c.init_hidden(batch_size) for c in self.encoding_cells], dim=0)


class DCGRUDecoder(_FusedDropout, nn.Module):
    """reference: model.py:112-204.  Time-major autoregressive loop; layers >= 1 share ONE cell
    object (model.py:126-143), so `decoding_cells.1` and `.2` alias the same parameters."""

    def __init__(self, input_dim, max_diffusion_step, num_nodes, hid_dim, output_dim, num_rnn_layers,
                 dcgru_activation=None, filter_type="laplacian", device=None, dropout=0.0):
        super().__init__()
        self.input_dim = input_dim
        self.hid_dim = hid_dim
        self.num_nodes = num_nodes
        self.output_dim = output_dim
        self.num_rnn_layers = num_rnn_layers
        self.max_diffusion_step = max_diffusion_step
        self._device = device
        shared = DCGRUCell(input_dim=hid_dim, num_units=hid_dim, max_diffusion_step=max_diffusion_step,
                           num_nodes=num_nodes, nonlinearity=dcgru_activation, filter_type=filter_type)
        first = DCGRUCell(input_dim=input_dim, num_units=hid_dim, max_diffusion_step=max_diffusion_step,
                          num_nodes=num_nodes, nonlinearity=dcgru_activation, filter_type=filter_type)
        self.decoding_cells = nn.ModuleList([first] + [shared] * (num_rnn_layers - 1))
        self.projection_layer = nn.Linear(hid_dim, output_dim)
        self.dropout = nn.Dropout(p=dropout)
        self._init_dropout_rng(1)

    def forward(self, inputs, initial_hidden_state, supports, teacher_forcing_ratio=None, teacher_flags=None):
        """inputs (T,B,N,Dout) targets, initial_hidden_state (L,B,N*H) -> (T,B,N*Dout).
        teacher_flags (extension): a DEVICE int32[T] tensor of teacher-forcing flags (`ops.teacher_flags`) instead of the
        host-side coin flips -- the persistent kernels read it when they start, so a captured HIP graph replays curriculum
        learning with a fresh draw every step.

        One native operator (eeg_dcrnn_decoder_fwd/bwd) runs the T autoregressive steps, the cells of
        all layers, the dropout in front of the projection (training, p > 0: a fresh mask per step, generated
        inside the persistent kernel) and the projection; the teacher-forcing coin flips (model.py:194-200: one
        `random.random()` per step) are drawn here, in the reference's order."""
        t_len, b = inputs.shape[0], inputs.shape[1]
        self.decoding_cells[0]._check_supports(supports)
        p, p_batched = ops.hop_polys(supports, self.max_diffusion_step, b)
        drop_p = self._drop_p()
        teacher = teacher_flags
        if teacher is None and teacher_forcing_ratio is not None:
            teacher = tuple(random.random() < teacher_forcing_ratio for _ in range(t_len))
        first = self.decoding_cells[0]
        shared = self.decoding_cells[1] if self.num_rnn_layers > 1 else None
        cell_params = lambda c: (c.dconv_gate.weight, c.dconv_gate.biases, c.dconv_candidate.weight,   # noqa: E731
                                 c.dconv_candidate.biases)
        return ops.dcgru_decoder(inputs.reshape(t_len, b, -1), initial_hidden_state, p, p_batched, cell_params(first),
                                 None if shared is None else cell_params(shared), self.projection_layer.weight,
                                 self.projection_layer.bias, self.num_nodes, self.hid_dim, self.output_dim,
                                 first.num_matrices, self.num_rnn_layers, first._activation_name, teacher,
                                 dropout_p=drop_p, rng_state=self._rng_state(inputs.device) if drop_p > 0 else None)


class DCRNNModel_classification(_FusedDropout, nn.Module):
    """Seizure detection / classification model (reference: model.py:208-272).
    forward(input_seq (B,T,N,Din), seq_lengths (B,), supports) -> (B, num_classes) logits."""

    def __init__(self, args, num_classes, device=None, strict_lengths=False):
        """strict_lengths: check `seq_lengths` on the host like the reference does (`utils.last_relevant_pytorch` moves them to
        the CPU and gathers at len-1: a length outside 1..T raises there, utils.py:346-357).  Off by default: the check costs the
        device-to-host synchronisation the reference pays every step; without it an out-of-range length selects the clamped step,
        in forward and backward alike."""
        super().__init__()
        self.num_nodes = args.num_nodes
        self.num_rnn_layers = args.num_rnn_layers
        self.rnn_units = args.rnn_units
        self._device = device
        self.num_classes = num_classes
        self.strict_lengths = bool(strict_lengths)
        self.encoder = DCRNNEncoder(input_dim=args.input_dim, max_diffusion_step=args.max_diffusion_step,
                                    hid_dim=args.rnn_units, num_nodes=args.num_nodes,
                                    num_rnn_layers=args.num_rnn_layers,
                                    dcgru_activation=args.dcgru_activation, filter_type=args.filter_type)
        self.fc = nn.Linear(args.rnn_units, num_classes)
        self.dropout = nn.Dropout(args.dropout)
        self.relu = nn.ReLU()
        self._init_dropout_rng(0)

    def encode_last(self, input_seq, seq_lengths, supports):
        """model.py:253-265: the top layer's state at t = seq_lengths-1 as (B, num_nodes, rnn_units) -- the input of the head"""
        b = input_seq.shape[0]
        if self.strict_lengths:
            utils.check_seq_lengths(seq_lengths, input_seq.shape[1])
        x = input_seq.transpose(0, 1)                         # (T,B,N,Din); made contiguous by the op
        _, _, last = self.encoder.run(x, None, supports, lengths=seq_lengths, want_finals=False)
        return last.view(b, self.num_nodes, self.rnn_units)

    def forward(self, input_seq, seq_lengths, supports):
        b = input_seq.shape[0]
        last = self.encode_last(input_seq, seq_lengths, supports)
        drop_p = self._drop_p()
        # dropout -> relu -> fc -> max over nodes in ONE launch (the mask is generated in the kernel, recomputed in its backward)
        return ops.cls_head(last.view(b, self.num_nodes, self.rnn_units), self.fc.weight, self.fc.bias, drop_p,
                            self._rng_state(last.device) if drop_p > 0 else None)


class DCRNNModel_nextTimePred(nn.Module):
    """Self-supervised next-clip prediction model (reference: model.py:277-360).
    forward(encoder_inputs (B,T,N,Din), decoder_inputs (B,T_out,N,Dout), supports, batches_seen)
    -> (B,T_out,N,Dout)."""

    def __init__(self, args, device=None):
        super().__init__()
        self.num_nodes = args.num_nodes
        self.num_rnn_layers = args.num_rnn_layers
        self.rnn_units = args.rnn_units
        self._device = device
        self.output_dim = args.output_dim
        self.cl_decay_steps = args.cl_decay_steps
        self.use_curriculum_learning = bool(args.use_curriculum_learning)
        # device-side scheduled sampling (see forward): what one forward adds to a `batches_seen` counter tensor
        self.batches_seen_increment = 0
        self._ddp_params_and_buffers_to_ignore = ["decoder._dropout_rng"]     # (see _FusedDropout._init_dropout_rng)
        self.encoder = DCRNNEncoder(input_dim=args.input_dim, max_diffusion_step=args.max_diffusion_step,
                                    hid_dim=args.rnn_units, num_nodes=args.num_nodes,
                                    num_rnn_layers=args.num_rnn_layers,
                                    dcgru_activation=args.dcgru_activation, filter_type=args.filter_type)
        self.decoder = DCGRUDecoder(input_dim=args.output_dim, max_diffusion_step=args.max_diffusion_step,
                                    num_nodes=args.num_nodes, hid_dim=args.rnn_units,
                                    output_dim=args.output_dim, num_rnn_layers=args.num_rnn_layers,
                                    dcgru_activation=args.dcgru_activation, filter_type=args.filter_type,
                                    device=device, dropout=args.dropout)

    def forward(self, encoder_inputs, decoder_inputs, supports, batches_seen=None):
        """batches_seen: the reference's host integer (model.py:336-343: the threshold and `random.random()` per decoder step
        are evaluated on the host) -- or, as an extension, a DEVICE int64[1] counter tensor: threshold and coin flips are then
        evaluated by `eeg_dcrnn_teacher_flags` on the stream from the decoder's Philox generator, and the counter advances
        by `self.batches_seen_increment` (train_ssl.py:178 `step += batch_size`), so that the forward is graph-replayable."""
        b, t_out, n, _ = decoder_inputs.shape
        enc_in = encoder_inputs.transpose(0, 1)
        dec_in = decoder_inputs.transpose(0, 1)
        enc_final, _, _ = self.encoder.run(enc_in, None, supports)
        ratio, flags = None, None
        if self.training and self.use_curriculum_learning and batches_seen is not None:
            if torch.is_tensor(batches_seen):
                flags = ops.teacher_flags(self.decoder._rng_state(batches_seen.device), batches_seen,
                                          self.batches_seen_increment, self.cl_decay_steps, t_out)
            else:
                ratio = utils.compute_sampling_threshold(self.cl_decay_steps, batches_seen)
        out = self.decoder(dec_in, enc_final, supports, teacher_forcing_ratio=ratio, teacher_flags=flags)
        return out.reshape(t_out, b, n, -1).transpose(0, 1)
