"""The reference's supervised/SSL optimisation step (train.py:222-224,253-275; train_ssl.py:147-178)
re-hosted for one-process-per-GPU data parallelism over RCCL.

Recipe per step:  zero_grad -> forward -> loss -> backward -> [all-reduce of ONE flat fp32
gradient bucket, mean over ranks] -> clip_grad_norm_(5.0) -> Adam(lr, weight_decay = coupled L2).

MI355X notes: the whole model is <= 2.8 MB, so parameters and gradients live in two flat
buffers (every `p` / `p.grad` is a view): the exchange is a single latency-bound all-reduce over
xGMI, the norm is one reduction, Adam runs over one tensor, and nothing in the step synchronises
with the host (the reference calls `loss.item()` and `lengths.cpu()` every step)."""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.distributed as dist

from . import ops


class FlatParameters:
    """Re-home a module's parameters (and their .grad) into two contiguous fp32 buffers.
    Shared parameters (the decoder's shared cell, model.py:126-143) appear once."""

    def __init__(self, module: torch.nn.Module):
        params, seen = [], set()
        for p in module.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        self.params = params
        total = sum(p.numel() for p in params)
        dev, dt = params[0].device, params[0].dtype
        self.flat = torch.empty(total, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(total, device=dev, dtype=dt)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                p.grad = self.flat_grad[off:off + n].view(p.shape)
                off += n
        self.flat_param = torch.nn.Parameter(self.flat, requires_grad=True)
        self.flat_param.grad = self.flat_grad
        # the backward operators write these parameters' gradients straight into the bucket while `with self.sink:` is open
        self.sink = ops.GradSink(params)

    def zero_grad(self):
        ops.zero_(self.flat_grad)          # one memset node on the stream (no framework fill kernel in the captured step)


class TrainStep:
    """One optimisation step of the reference recipe on the local shard of the global batch."""

    def __init__(self, model: torch.nn.Module, task: str = "detection", lr: float = 3e-4,
                 weight_decay: float = 5e-4, max_grad_norm: float = 5.0,
                 scaler_mean: Optional[float] = None, scaler_std: Optional[float] = None,
                 always_reduce: bool = False, raw_window: Optional[int] = None, raw_mean: float = 0.0, raw_std: float = 1.0,
                 shared_graph: bool = True, data_augment: bool = False, swap_perm=None, reflected_supports=None,
                 feature_std: Optional[float] = None):
        """shared_graph: batched supports whose clips all carry one and the same graph (the reference's trainers pass the distance
        graph that way) are handed to the model in their 2-D form, which lets the encoder run its hoisted GEMMs in the
        eigenbasis of that graph (`ops.collapse_shared_supports`: one comparison + one flag read per supports TENSOR, cached; a
        captured step keeps the decision of its capture -- refill a captured supports buffer with per-clip graphs only after
        re-capturing, or pass shared_graph=False).
        always_reduce: issue the gradient all-reduce whenever a process group exists, also at world size 1 (exercises
        the RCCL path on a single GPU; a sum over one rank is the identity).
        data_augment: the reference's training-set augmentation (`--data_augment`; dataloader_detection.py:233-256,384-389: per
        sample a fair coin -- reflect the electrodes along the midline or not -- and an amplitude factor uniform in [0.8, 1.2), which
        under use_fft is `+= log(factor)` before standardisation), drawn ON THE DEVICE every step (`ops.draw_augmentation`, a Philox
        state of this object: a captured step draws afresh at every replay).  With raw_window the draws are operands of the
        featurisation kernel; for feature inputs (already standardised) the step applies x[b, :, perm[b]] + log(factor_b) /
        feature_std, which is the same value (feature_std = the scaler's std, required then).  Graph side as in the reference:
        supports=None (correlation graph) is built from the UN-reflected, un-scaled clip (`_get_indiv_graphs` never reads its
        swapped name table); for the distance graph pass `reflected_supports` (`utils.reflected_supports`: the graph of
        `_get_combined_graph(swap_nodes)`) -- every clip then carries the plain or the reflected supports according to its coin
        (per-clip graphs: the general path, not the spectral form).  swap_perm: `utils.swap_permutation(num_nodes)` by default.
        raw_window: the step takes RAW resampled signals (B, N, T*raw_window) instead of features and runs the reference's
        DataLoader-side chain on the device in front of the model (dataloader_detection.py:57-71,346-354,384-393): log|FFT| of every
        raw_window-sample step (`eeg_dcrnn_fft_features`) -> z-score with (raw_mean, raw_std) = the model input; with
        supports=None the per-clip correlation graph is built from the UN-standardised features, as `_get_indiv_graphs` does."""
        assert task in ("detection", "classification", "ssl")
        self.model, self.task, self.max_grad_norm = model, task, max_grad_norm
        self.fp = FlatParameters(model)
        # optimiser state lives in flat buffers; the update is ONE fused HIP kernel (clip + Adam)
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, (0.9, 0.999), 1e-8
        self.exp_avg = torch.zeros_like(self.fp.flat)
        self.exp_avg_sq = torch.zeros_like(self.fp.flat)
        self.ws = torch.zeros(64, device=self.fp.flat.device, dtype=torch.float32)
        self.grad_norm = torch.zeros(1, device=self.fp.flat.device, dtype=torch.float32)
        self.step_count = 0
        # the optimiser's step count and learning rate also live on the device (eeg_dcrnn_clip_adam_dev reads / advances them on
        # the stream): the update is then a graph node like everything else.  `step_count` / `lr` are the host mirrors.
        dev = self.fp.flat.device
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.lr_dev = torch.full((1,), float(lr), device=dev, dtype=torch.float32)
        # samples seen so far = the reference's `step += batch_size` (train_ssl.py:163,178): drives the scheduled-
        # sampling threshold of the SSL model under curriculum learning (global batch: every rank advances alike)
        self.samples_seen = 0
        self.samples_seen_dev = torch.zeros(1, device=dev, dtype=torch.int64)
        has_pg = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if has_pg else 1
        self.reduce = has_pg and (self.world > 1 or always_reduce)
        self.scaler_mean, self.scaler_std = scaler_mean, scaler_std
        self.raw_window, self.raw_mean, self.raw_std = raw_window, float(raw_mean), float(raw_std)
        self.shared_graph = bool(shared_graph)
        # detection / classification: head + criterion + head backward as ONE operator behind the encoder (ops.cls_head_loss) instead
        # of model.forward -> loss kernel -> autograd through the head (same values; False: that public path, launch by launch)
        self.fused_head = True
        self.last_logits = None
        self.data_augment = bool(data_augment)
        self.feature_std = None if feature_std is None else float(feature_std)
        self.swap_perm, self.reflected_supports, self._augment_rng = None, None, None
        if self.data_augment:
            from . import utils
            if raw_window is None and feature_std is None:
                raise ValueError("TrainStep: data_augment on feature inputs needs feature_std (the StandardScaler's std: the reference "
                                 "adds log(scale) BEFORE it standardises)")
            sp = utils.swap_permutation(model.num_nodes) if swap_perm is None else torch.as_tensor(swap_perm)
            if sp.numel() != model.num_nodes:
                raise ValueError(f"TrainStep: swap_perm has {sp.numel()} entries for {model.num_nodes} nodes")
            self.swap_perm = sp.to(device=dev, dtype=torch.int32).contiguous()
            if reflected_supports is not None:
                self.reflected_supports = torch.stack([torch.as_tensor(r, dtype=torch.float32) for r in reflected_supports]).to(dev).contiguous()
            self._augment_rng = ops.make_rng_state(dev, stream_id=2)
        self.last_augmentation = None     # (flags, perm, log_scale) of the latest step (tests / logging)
        self._graphs = {}
        # curriculum learning (SSL, model.py:194-200): where the persistent decoder kernels apply, the teacher-forcing flags are
        # drawn on the device (`eeg_dcrnn_teacher_flags`) -- in eager steps and graph replays alike; elsewhere on the host
        self.device_curriculum = None     # decided at the first batch (needs the shapes)

    # `step_count` / `samples_seen`: host mirrors of the device-resident counters the kernels read (Adam's bias correction uses
    # step_dev, the scheduled-sampling threshold samples_seen_dev); assigning to them (a resumed or restarted run) rewrites the
    # device tensors too -- outside any captured graph, like `lr`.
    @property
    def step_count(self):
        return self._step_count

    @step_count.setter
    def step_count(self, value):
        self._step_count = int(value)
        if hasattr(self, "step_dev"):
            self.step_dev.fill_(self._step_count)

    @property
    def samples_seen(self):
        return self._samples_seen

    @samples_seen.setter
    def samples_seen(self, value):
        self._samples_seen = int(value)
        if hasattr(self, "samples_seen_dev"):
            self.samples_seen_dev.fill_(self._samples_seen)

    def _advance(self, steps: int = 0, samples: int = 0):
        """the kernels of a step advanced the device counters themselves: move the host mirrors only"""
        self._step_count += int(steps)
        self._samples_seen += int(samples)

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        if hasattr(self, "lr_dev"):
            self.lr_dev.fill_(self._lr)   # (outside any captured graph: the graph reads the tensor)

    def _use_device_curriculum(self, y) -> bool:
        m = self.model
        if self.task != "ssl" or not getattr(m, "use_curriculum_learning", False):
            return False
        if self.device_curriculum is None:
            dec = m.decoder
            self.device_curriculum = ops.decoder_is_persistent(y.shape[1], y.shape[0], dec.num_nodes, dec.hid_dim, dec.output_dim,
                                                               dec.decoding_cells[0].num_matrices, dec.num_rnn_layers)
        return self.device_curriculum

    def set_epoch(self, epoch: int, num_epochs: int, eta_min: float = 0.0):
        """Cosine learning-rate schedule of the reference, stepped per epoch (train.py:224,329)."""
        from . import utils
        if not hasattr(self, "base_lr"):
            self.base_lr = self.lr
        self.lr = utils.cosine_annealing_lr(self.base_lr, epoch, num_epochs, eta_min)
        return self.lr

    def loss(self, out, y):
        if self.task == "detection":        # train.py:203-204,266-267
            return ops.bce_with_logits(out.view(-1), y)
        if self.task == "classification":   # train.py:205-206,268
            return ops.cross_entropy(out, y)
        from . import utils                 # train_ssl.py:165-170 ("MAE" -> masked RMSE, Q9)
        sc = None if self.scaler_mean is None else utils.StandardScaler(self.scaler_mean, self.scaler_std)
        return utils.compute_regression_loss(y_true=y, y_predicted=out, standard_scaler=sc, loss_fn="MAE")

    def loss_and_grad(self, out, y):
        """(loss, d loss / d out) of the task's criterion from the fused HIP loss kernels (same values as `self.loss`)"""
        if self.task == "detection":
            return torch.ops.eeg_dcrnn.bce_logits(out.view(-1), y)
        if self.task == "classification":
            return torch.ops.eeg_dcrnn.ce_logits(out, y)
        scaled = self.scaler_mean is not None
        # train_ssl.py:165-170: loss_fn "MAE" != "mae" selects the masked RMSE (kind 1), Q9
        return torch.ops.eeg_dcrnn.masked_loss(out, y, scaled, float(self.scaler_mean) if scaled else 0.0,
                                               float(self.scaler_std) if scaled else 1.0, 0.0, 1)

    def forward_backward(self, x, y, seq_lengths, supports):
        """supports=None: build the per-clip correlation graph and its dual random-walk supports from
        the clips on the device (the DataLoader-side `_get_indiv_graphs` of the reference)."""
        self.fp.zero_grad()
        perm, log_scale = None, None
        if self.data_augment and self.model.training:
            supports, perm, log_scale = self._draw_augmentation(x.shape[0], supports)
        if self.raw_window is not None:              # raw signals in: featurise on the device (x becomes the standardised log|FFT|)
            feat_raw, x = ops.fft_features(x, window=self.raw_window, mean=self.raw_mean, std=self.raw_std, perm=perm, log_scale=log_scale)
            if supports is None:
                supports = ops.correlation_supports(feat_raw, top_k=3)       # (feat_raw: un-reflected, un-scaled, un-standardised)
        elif perm is not None:
            plain = x
            idx = perm.to(torch.int64)[:, None, :, None].expand(-1, x.shape[1], -1, x.shape[3])
            x = x.gather(2, idx) + (log_scale / self.feature_std)[:, None, None, None]
            if supports is None:
                supports = ops.correlation_supports(plain, top_k=3)
        if supports is None:
            supports = ops.correlation_supports(x, top_k=3)
        elif self.shared_graph:
            supports = ops.collapse_shared_supports(supports)
        if self.task == "ssl":
            if self._use_device_curriculum(y):
                self.model.batches_seen_increment = x.shape[0] * self.world
                out = self.model(x, y, supports, batches_seen=self.samples_seen_dev)
            else:
                out = self.model(x, y, supports, batches_seen=self.samples_seen)    # train_ssl.py:163
        elif self.fused_head and hasattr(self.model, "encode_last"):
            # encoder -> [head, criterion and the head's backward: two launches] -> encoder backward seeded with d loss / d last
            m = self.model
            last = m.encode_last(x, seq_lengths, supports)
            drop_p = m._drop_p()
            with self.fp.sink:
                loss, self.last_logits, dz = ops.cls_head_loss(last, m.fc.weight, m.fc.bias, y, self.task, drop_p,
                                                               m._rng_state(last.device) if drop_p > 0 else None)
                last.backward(dz)
            return loss.detach()
        else:
            out = self.model(x, seq_lengths, supports)
        # The loss kernels return value AND gradient (d loss / d out) from one pass: backward is seeded with that gradient
        # directly -- `loss.backward()` would first fill a ones tensor and multiply the saved gradient by it (two framework
        # kernels per step for a factor of exactly 1).  `self.loss(out, y).backward()` remains the equivalent public path.
        loss, seed = self.loss_and_grad(out.detach(), y)
        with self.fp.sink:                       # backward operators write into the flat gradient bucket
            out.backward(seed.view_as(out))
        return loss.detach()

    def _draw_augmentation(self, batch, supports):
        """this step's draws; with a distance graph and its reflected partner, the per-clip supports"""
        plain = None
        if supports is not None and self.reflected_supports is not None:
            shared = ops.collapse_shared_supports(supports) if self.shared_graph else supports
            if any(s_.dim() != 2 for s_ in shared):
                raise RuntimeError("TrainStep(data_augment, reflected_supports): the supports of the step must be ONE graph shared by all "
                                   "clips (2-D tensors, or batched copies of it): the reflected partner is chosen per clip")
            if len(shared) != self.reflected_supports.shape[0]:
                raise RuntimeError(f"TrainStep: {len(shared)} supports, {self.reflected_supports.shape[0]} reflected partners")
            plain = torch.stack([s_.to(torch.float32) for s_ in shared])
        flags, perm, log_scale, sel = ops.draw_augmentation(self._augment_rng, batch, self.swap_perm, plain,
                                                            None if plain is None else self.reflected_supports)
        self.last_augmentation = (flags, perm, log_scale)
        return (supports if sel is None else sel), perm, log_scale

    # -- HIP-graph replay of forward + loss + backward -------------------------------------------
    def capture(self, x, y, seq_lengths, supports, warmup: int = 2, slot: int = 0, include_update: bool = False):
        """include_update: also capture the exchange + optimiser tail (the RCCL all-reduce of the flat bucket when a process
        group exists -- RCCL collectives are capturable -- and the fused clip + Adam, whose step count and learning rate are
        device-resident): the WHOLE optimisation step is then one graph launch per rank.  The warm-up launches and the upload
        replay below then apply real updates; a caller that minds restores the state afterwards (`snapshot()` / `restore()`).

        Capture zero_grad -> forward -> loss -> backward on the given (static) input tensors into
        one HIP graph (torch.cuda.CUDAGraph: the library launches on torch's current stream, never
        synchronises and allocates only through torch, so the ~60 launches of a step replay as one
        graph launch).  The exchange + optimiser tail stay outside the graph: the RCCL all-reduce is
        issued eagerly between the replay and the fused clip+Adam kernel.  New data is fed by
        copying into the captured tensors (`x.copy_(batch)`) -- or, without any device-side copy, by capturing the step
        on TWO input sets (`slot` 0 and 1) and alternating `replay_step(slot)`: the host-to-device copy of batch k+1
        then lands directly in the tensors the next replay reads while batch k computes."""
        if self.task == "ssl" and getattr(self.model, "use_curriculum_learning", False) and not self._use_device_curriculum(y):
            # host-side teacher-forcing coin flips (model.py:194-200) select which launches are issued: a captured graph would
            # freeze one draw for ever.  (Where the persistent decoder kernels apply the flags are drawn on the device instead.)
            raise RuntimeError("TrainStep.capture: this decoder shape is outside the persistent decoder kernels, so curriculum "
                               "learning draws its teacher-forcing flags on the host every step; use step() (eager launches)")
        keep = self.snapshot() if self.task == "ssl" and self._use_device_curriculum(y) and not include_update else None

        def body():
            loss = self.forward_backward(x, y, seq_lengths, supports)
            if include_update:
                self.reduce_and_update(count=False)
            return loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                       # populate the allocator before capture
                body()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = body()
        # one untimed replay: the first launch of an instantiated graph also uploads it to the device (torch exposes no
        # hipGraphUpload); it recomputes the gradients of the captured batch (and, with include_update, applies one more update)
        graph.replay()
        if include_update:
            self._advance(warmup + 1, (warmup + 1) * x.shape[0] * self.world)
        elif keep is not None:
            self.restore(keep, counters_only=True)        # the warm-up draws advanced the device-side sample counter
        self._graphs[slot] = (graph, loss, (x, y, seq_lengths, supports), include_update)
        return graph

    def snapshot(self):
        """everything a step changes besides the gradients: parameters, Adam moments, counters (device and host)"""
        return {"flat": self.fp.flat.detach().clone(), "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "step_dev": self.step_dev.clone(), "samples_seen_dev": self.samples_seen_dev.clone(),
                "step_count": self.step_count, "samples_seen": self.samples_seen}

    def restore(self, snap, counters_only: bool = False):
        with torch.no_grad():
            if not counters_only:
                self.fp.flat.copy_(snap["flat"])
                self.exp_avg.copy_(snap["exp_avg"])
                self.exp_avg_sq.copy_(snap["exp_avg_sq"])
                self.step_dev.copy_(snap["step_dev"])
                self.step_count = snap["step_count"]
            self.samples_seen_dev.copy_(snap["samples_seen_dev"])
            self.samples_seen = snap["samples_seen"]

    def replay_step(self, slot: int = 0):
        """One optimisation step on the captured tensors of `slot`: graph replay + all-reduce + clip/Adam."""
        graph, loss, inputs, whole = self._graphs[slot]
        graph.replay()
        self._advance(0, inputs[0].shape[0] * self.world)
        if whole:
            self._advance(1)
        else:
            self.reduce_and_update()
        return loss

    def reduce_and_update(self, count: bool = True):
        g = self.fp.flat_grad
        if self.reduce:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)        # RCCL over xGMI: one flat bucket
        if count:
            self._advance(1)
        # mean over ranks (grad_scale), clip_grad_norm_(max_norm) and Adam in one pass over the buffers; the kernel advances
        # the device-resident step count itself
        ops.clip_adam_step_dev(self.fp.flat, g, self.exp_avg, self.exp_avg_sq, self.step_dev, self.lr_dev, self.betas,
                               self.eps, self.weight_decay, self.max_grad_norm, 1.0 / self.world, self.ws, self.grad_norm)
        return self.grad_norm

    def step(self, x, y, seq_lengths, supports):
        loss = self.forward_backward(x, y, seq_lengths, supports)
        self._advance(0, x.shape[0] * self.world)
        self.reduce_and_update()
        return loss

    # -- checkpointing (utils.CheckpointSaver / load_model_checkpoint use these like an optimizer's) -------
    def state_dict(self):
        return {"step": self.step_count, "samples_seen": self.samples_seen, "lr": self.lr, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone()}

    def load_state_dict(self, state):
        self.step_count, self.lr = int(state["step"]), float(state["lr"])
        self.samples_seen = int(state.get("samples_seen", 0))
        self.step_dev.fill_(self.step_count)
        self.samples_seen_dev.fill_(self.samples_seen)
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])


def _all_gather_uneven(t: torch.Tensor) -> torch.Tensor:
    """Concatenation over the ranks of tensors whose FIRST dimension differs from rank to rank (evaluation shards of a
    data set are uneven whenever its size is not a multiple of the world size): the lengths travel first, every rank
    pads its shard to the longest one, one all_gather, and the padding is cut away again.  Same result on every rank."""
    world = dist.get_world_size()
    n = torch.tensor([t.shape[0]], device=t.device, dtype=torch.int64)
    sizes = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    longest = max(sizes)
    if t.shape[0] < longest:
        t = torch.cat([t, t.new_zeros((longest - t.shape[0],) + tuple(t.shape[1:]))])
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous())
    return torch.cat([p[:k] for p, k in zip(parts, sizes)])


@torch.no_grad()
def predict(model, batches, task: str = "detection"):
    """Evaluation forward passes (train.py:343-404 without the host round trip per batch): returns
    (y_prob, y_true) as numpy arrays gathered from ALL ranks on every rank (RCCL all_gather when launched
    data-parallel).  batches: iterable of (x, y, seq_lengths, supports) device tensors."""
    was_training = model.training
    model.eval()
    probs, labels = [], []
    for x, y, seq_lengths, supports in batches:
        if supports is None:
            supports = ops.correlation_supports(x, top_k=3)
        logits = model(x, seq_lengths, supports)
        probs.append(torch.sigmoid(logits.view(-1)) if task == "detection" else torch.softmax(logits, dim=1))
        labels.append(y.view(-1))
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if probs:
        prob, lab = torch.cat(probs), torch.cat(labels)
    elif multi:
        # a rank whose shard of the evaluation set is empty still takes part in the gather (the others are waiting in it)
        dev = next(model.parameters()).device
        prob = torch.empty((0,) if task == "detection" else (0, model.fc.out_features), device=dev)
        lab = torch.empty((0,), dtype=torch.float32 if task == "detection" else torch.int64, device=dev)
    else:
        model.train(was_training)
        raise ValueError("predict: no batches")
    if multi:
        prob, lab = _all_gather_uneven(prob), _all_gather_uneven(lab)
    model.train(was_training)
    return prob.cpu().numpy(), lab.cpu().numpy()


@torch.no_grad()
def evaluate_ssl(model, batches, scaler_mean: Optional[float] = None, scaler_std: Optional[float] = None,
                 return_predictions: bool = False):
    """The reference's SSL evaluation pass (train_ssl.py:232-280): eval mode (no dropout, no teacher forcing -- `model(x, y,
    supports)` without `batches_seen`), per batch the masked MAE in original units (`loss_fn="mae"` with the StandardScaler,
    utils.py:431-495) from the HIP loss kernel, averaged over the data set weighted by batch size (the reference's
    `AverageMeter`, train_ssl.py:262-263,276).  Launched data-parallel, every rank evaluates its shard and all ranks
    return the loss of the union (all-reduce of the weighted sum and the count; no host round trip per batch).
    batches: iterable of (x, y, supports) or (x, y, seq_lengths, supports) device tensors (supports None: built on the
    device).  Returns eval_loss (float) -- and, with return_predictions, the predictions and targets of THIS rank's shard
    as the reference collects them (train_ssl.py:266-274)."""
    was_training = model.training
    model.eval()
    tot, preds, truths = None, [], []
    for batch in batches:
        x, y, supports = batch[0], batch[1], batch[-1]
        if supports is None:
            supports = ops.correlation_supports(x, top_k=3)
        pred = model(x, y, supports)
        loss = ops.masked_regression_loss(pred, y, scaler_mean, scaler_std, loss_fn="mae")
        w = torch.stack([loss.reshape(()).double() * x.shape[0], torch.tensor(float(x.shape[0]), device=x.device, dtype=torch.float64)])
        tot = w if tot is None else tot + w
        if return_predictions:
            preds.append(pred)
            truths.append(y)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if tot is None:
        if not multi:
            model.train(was_training)
            raise ValueError("evaluate_ssl: no batches")
        tot = torch.zeros(2, dtype=torch.float64, device=next(model.parameters()).device)    # an empty shard still joins the all-reduce
    if multi:
        dist.all_reduce(tot)
    model.train(was_training)
    if float(tot[1].item()) == 0.0:
        raise ValueError("evaluate_ssl: no batches on any rank")
    eval_loss = float((tot[0] / tot[1]).item())
    if return_predictions:
        return eval_loss, torch.cat(preds).cpu().numpy(), torch.cat(truths).cpu().numpy()
    return eval_loss


@torch.no_grad()
def evaluate(model, batches, task: str = "detection", is_test: bool = False, eval_set: str = "dev",
             best_thresh: float = 0.5):
    """The reference's evaluation pass (train.py:332-431) on device tensors: forward in eval mode, sample-weighted
    mean loss (BCE-with-logits / cross-entropy, computed by the HIP loss kernels), predictions at `best_thresh`
    (detection; on the dev set of a test run the threshold is re-chosen by `utils.thresh_max_f1`,
    train.py:406-412) or arg-max (classification), and the score dictionary in the reference's order:
    loss, acc, F1, recall, precision, best_thresh[, auroc].  Launched data-parallel, every rank evaluates its
    shard and all ranks return the scores of the union (all_gather of probabilities, labels and loss sums).
    batches: iterable of (x, y, seq_lengths, supports) device tensors (supports None: built on the device)."""
    from collections import OrderedDict
    import numpy as np
    from . import utils
    was_training = model.training
    model.eval()
    probs, labels = [], []
    loss_sum = None
    n_seen = 0
    for x, y, seq_lengths, supports in batches:
        if supports is None:
            supports = ops.correlation_supports(x, top_k=3)
        logits = model(x, seq_lengths, supports)
        if task == "detection":
            lg = logits.view(-1)
            loss = ops.bce_with_logits(lg, y.view(-1).float())
            probs.append(torch.sigmoid(lg))
        else:
            loss = ops.cross_entropy(logits, y.view(-1))
            probs.append(torch.softmax(logits, dim=1))
        labels.append(y.view(-1))
        loss_sum = loss * x.shape[0] if loss_sum is None else loss_sum + loss * x.shape[0]
        n_seen += x.shape[0]
    prob, lab = torch.cat(probs), torch.cat(labels)
    tot = torch.stack([loss_sum.reshape(()).double(), torch.tensor(float(n_seen), device=prob.device, dtype=torch.float64)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        prob, lab = _all_gather_uneven(prob), _all_gather_uneven(lab)
        dist.all_reduce(tot)
    model.train(was_training)
    y_prob, y_true = prob.cpu().numpy(), lab.cpu().numpy().astype(int)
    eval_loss = float((tot[0] / tot[1]).item())
    if task == "detection":
        if eval_set == "dev" and is_test:
            best_thresh = float(utils.thresh_max_f1(y_true=y_true, y_prob=y_prob))
        y_pred = (y_prob > best_thresh).astype(int)
    else:
        y_pred = np.argmax(y_prob, axis=1).reshape(-1)
    scores, _, _ = utils.eval_dict(y_pred=y_pred, y=y_true, y_prob=y_prob if task == "detection" else None,
                                   average="binary" if task == "detection" else "weighted")
    res = [("loss", eval_loss), ("acc", scores["acc"]), ("F1", scores["F1"]), ("recall", scores["recall"]),
           ("precision", scores["precision"]), ("best_thresh", best_thresh)]
    if "auroc" in scores:
        res.append(("auroc", scores["auroc"]))
    return OrderedDict(res)
