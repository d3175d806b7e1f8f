"""Hot-path helpers with the reference's names (tsy935/eeg-gnn-ssl utils.py / data_utils.py):
graph supports that feed the diffusion convolution, the gather at len-1, the losses that seed
backward, and the checkpoint transplant used by fine-tuning.  Only what the DCRNN path needs."""
import math

import numpy as np
import torch

from . import ops


# ---- supports (host side, float64 like the reference's scipy path) ----------------------------
def calculate_normalized_laplacian(adj):
    """L = I - D^-1/2 A D^-1/2 (reference utils.py:205-217); dense ndarray in, dense out."""
    adj = np.asarray(adj)
    deg = adj.sum(axis=1)
    with np.errstate(divide="ignore"):
        dis = np.power(deg, -0.5)
    dis[np.isinf(dis)] = 0.0
    return np.eye(adj.shape[0]) - (adj * dis[None, :]).T * dis[None, :]


def calculate_random_walk_matrix(adj_mx):
    """D_o^-1 W (reference utils.py:220-230)."""
    adj_mx = np.asarray(adj_mx)
    deg = adj_mx.sum(axis=1)
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -1.0)
    dinv[np.isinf(dinv)] = 0.0
    return dinv[:, None] * adj_mx


def calculate_reverse_random_walk_matrix(adj_mx):
    """D_i^-1 W^T (reference utils.py:233-237)."""
    return calculate_random_walk_matrix(np.transpose(adj_mx))


def calculate_scaled_laplacian(adj_mx, lambda_max=2, undirected=True):
    """2 L / lambda_max - I (reference utils.py:240-255); `lambda_max=None` -> spectral radius."""
    adj_mx = np.asarray(adj_mx)
    if undirected:
        adj_mx = np.maximum(adj_mx, adj_mx.T)
    lap = calculate_normalized_laplacian(adj_mx)
    if lambda_max is None:
        ev = np.linalg.eigvalsh((lap + lap.T) * 0.5)
        lambda_max = ev[np.argmax(np.abs(ev))]
    return (2.0 / lambda_max) * lap - np.eye(lap.shape[0])


def compute_supports(adj_mat, filter_type):
    """dataloader_detection.py:335-354 (`_compute_supports`): list of float32 (N,N) tensors."""
    if filter_type == "laplacian":
        mats = [calculate_scaled_laplacian(adj_mat, lambda_max=None)]
    elif filter_type == "random_walk":
        mats = [calculate_random_walk_matrix(adj_mat).T]
    elif filter_type == "dual_random_walk":
        mats = [calculate_random_walk_matrix(adj_mat).T, calculate_random_walk_matrix(np.transpose(adj_mat)).T]
    else:
        mats = [calculate_scaled_laplacian(adj_mat)]
    return [torch.from_numpy(np.ascontiguousarray(m)).to(torch.float32) for m in mats]


def keep_topk(adj_mat, top_k=3, directed=True):
    """data_utils.py:174-200: keep each node's top-k neighbours (plus the diagonal)."""
    work = np.array(adj_mat, copy=True)
    np.fill_diagonal(work, 0)
    nbr = np.argsort(-work, axis=-1)[:, :top_k]
    keep = np.eye(work.shape[0], dtype=bool)
    rows = np.repeat(np.arange(work.shape[0]), nbr.shape[1])
    keep[rows, nbr.reshape(-1)] = True
    if not directed:
        keep[nbr.reshape(-1), rows] = True
    return keep * adj_mat


def correlation_graph(clip, top_k=3):
    """Per-clip correlation adjacency (dataloader_detection.py:258-307): |cosine Gram| of the
    (N, T*D) clip, unit diagonal, top-k directed.  clip: (T, N, D) ndarray -> (N, N) float32."""
    n = clip.shape[1]
    flat = np.transpose(clip, (1, 0, 2)).reshape(n, -1).astype(np.float64)
    norm = np.sqrt((flat * flat).sum(axis=1))
    gram = flat @ flat.T
    denom = np.outer(norm, norm)
    adj = np.divide(gram, denom, out=gram.copy(), where=denom != 0).astype(np.float32)
    np.fill_diagonal(adj, 1.0)
    return keep_topk(np.abs(adj), top_k=top_k, directed=True)


# ---- reflection augmentation: the graph side (host, once per run) --------------------------------------------
# the 19-electrode montage in the reference's channel order (constants.py:2-21)
INCLUDED_CHANNELS = ["EEG FP1", "EEG FP2", "EEG F3", "EEG F4", "EEG C3", "EEG C4", "EEG P3", "EEG P4", "EEG O1", "EEG O2",
                     "EEG F7", "EEG F8", "EEG T3", "EEG T4", "EEG T5", "EEG T6", "EEG FZ", "EEG CZ", "EEG PZ"]
_MIDLINE_PAIRS = (("EEG FP1", "EEG FP2"), ("EEG Fp1", "EEG Fp2"), ("EEG F3", "EEG F4"), ("EEG F7", "EEG F8"),
                  ("EEG C3", "EEG C4"), ("EEG T3", "EEG T4"), ("EEG T5", "EEG T6"), ("EEG O1", "EEG O2"))


def get_swap_pairs(channels=None):
    """data_utils.py:37-62: index pairs mirrored along the midline by `_random_reflect`, in the reference's order (P3 / P4 are
    NOT among them -- kept as is)."""
    channels = INCLUDED_CHANNELS if channels is None else list(channels)
    return [(channels.index(a), channels.index(b)) for a, b in _MIDLINE_PAIRS if a in channels and b in channels]


def swap_permutation(num_nodes, swap_pairs=None):
    """source channel of every node of a reflected clip (`EEG_seq_reflect[:, [a, b]] = EEG_seq[:, [b, a]]`,
    dataloader_detection.py:233-246) as an int32 (N,) tensor -- the `swap_perm` of `ops.draw_augmentation`."""
    perm = np.arange(num_nodes, dtype=np.int32)
    for a, b in (get_swap_pairs() if swap_pairs is None else swap_pairs):
        perm[a], perm[b] = b, a
    return torch.from_numpy(perm)


def reflected_adjacency(adj_mat, swap_pairs=None):
    """`_get_combined_graph(swap_nodes)` (dataloader_detection.py:309-333): the distance-graph adjacency a REFLECTED clip is
    paired with.  Every assignment of the reference's loop reads the ORIGINAL matrix, so a position touched by two pairs keeps
    the later pair's value only: entry (a, c) of pairs (a, b), (c, d) ends as adj[a, d], not adj[b, d] -- the result is symmetric
    but NOT the permutation similarity P A P^T (its spectrum differs from the plain graph's).  Reproduced as is."""
    adj = np.asarray(adj_mat)
    new = adj.copy()
    for a, b in (get_swap_pairs() if swap_pairs is None else swap_pairs):
        new[[a, b], :] = adj[[b, a], :]
        new[:, [a, b]] = adj[:, [b, a]]
        np.fill_diagonal(new, 1)
        new[a, b], new[b, a] = adj[b, a], adj[a, b]
    return new


def reflected_supports(adj_mat, filter_type, swap_pairs=None):
    """supports of the reflected distance graph (`_compute_supports(_get_combined_graph(swap_nodes))`,
    dataloader_detection.py:405-409): list of float32 (N,N) tensors, the partner of `compute_supports(adj_mat, filter_type)`."""
    return compute_supports(reflected_adjacency(adj_mat, swap_pairs), filter_type)


# ---- sequence helpers -----------------------------------------------------------------------
def last_relevant_pytorch(output, lengths, batch_first=True):
    """Gather `output` at t = lengths-1 (reference utils.py:346-357).  Stays on the device (the
    reference forces `lengths.cpu()`).  Differentiable through a plain torch gather; the
    classification model uses the fused device path instead."""
    idx = (lengths.to(device=output.device, dtype=torch.int64) - 1).view(-1, 1, 1)
    if batch_first:
        return output.gather(1, idx.expand(-1, 1, output.size(2))).squeeze(1)
    return output.gather(0, idx.view(1, -1, 1).expand(1, -1, output.size(2))).squeeze(0)


def compute_sampling_threshold(cl_decay_steps, global_step):
    """Scheduled-sampling threshold (reference utils.py:385-390)."""
    return cl_decay_steps / (cl_decay_steps + math.exp(global_step / cl_decay_steps))


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


# ---- regression losses of the SSL task --------------------------------------------------------
class StandardScaler:
    """reference utils.py:393-428 with scalar (or broadcastable) mean / std."""

    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def transform(self, data):
        return (data - self.mean) / self.std

    def inverse_transform(self, data, is_tensor=False, device=None, mask=None):
        mean, std = self.mean, self.std
        if is_tensor:
            mean = torch.as_tensor(np.asarray(mean), dtype=torch.float32, device=data.device)
            std = torch.as_tensor(np.asarray(std), dtype=torch.float32, device=data.device)
        return data * std + mean


def _masked(y_pred, y_true, mask_val, elementwise):
    w = (y_true != mask_val).to(y_pred.dtype)
    w = w / w.mean()
    loss = elementwise(y_pred - y_true) * w
    return torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)


def masked_mae_loss(y_pred, y_true, mask_val=0.0):
    """reference utils.py:431-442."""
    return _masked(y_pred, y_true, mask_val, torch.abs).mean()


def masked_mse_loss(y_pred, y_true, mask_val=0.0):
    """reference utils.py:445-457 — despite the name this is the masked RMSE (sqrt of the mean)."""
    return torch.sqrt(_masked(y_pred, y_true, mask_val, lambda d: d * d).mean())


def check_seq_lengths(seq_lengths, max_len):
    """The reference's error behaviour for bad `seq_lengths` (utils.py:346-357: `output.gather(1, lengths - 1)` on host-side
    lengths raises for a length outside 1..T), as an explicit host-side check (one device-to-host synchronisation)."""
    ln = torch.as_tensor(seq_lengths).detach().to("cpu", torch.int64)
    bad = (ln < 1) | (ln > int(max_len))
    if bool(bad.any()):
        i = int(torch.nonzero(bad)[0])
        raise RuntimeError(f"index {int(ln[i]) - 1} is out of bounds for dimension 1 with size {int(max_len)} "
                           f"(seq_lengths[{i}] = {int(ln[i])}, valid: 1..{int(max_len)})")


def compute_regression_loss(y_true, y_predicted, standard_scaler=None, device=None, loss_fn="mae",
                            mask_val=0.0, is_tensor=True):
    """reference utils.py:460-495.  Only the exact string 'mae' selects the MAE; the SSL trainer
    passes "MAE" (train_ssl.py:168) and therefore optimises the masked RMSE — kept as is.

    Tensor inputs with a scalar scaler (the reference's case: utils.py:402-403 pickled scalars) run
    in the HIP loss kernels (value + gradient, eeg_dcrnn_masked_loss); numpy inputs
    (`is_tensor=False`) are host-side evaluation utilities."""
    if device is not None:
        y_true, y_predicted = y_true.to(device), y_predicted.to(device)
    scalar_scaler = standard_scaler is None or (np.ndim(standard_scaler.mean) == 0 and np.ndim(standard_scaler.std) == 0)
    if is_tensor and scalar_scaler and torch.is_tensor(y_predicted):
        from . import ops
        mean = None if standard_scaler is None else float(standard_scaler.mean)
        std = None if standard_scaler is None else float(standard_scaler.std)
        return ops.masked_regression_loss(y_predicted, y_true, mean, std, loss_fn, mask_val)
    if standard_scaler is not None:
        y_true = standard_scaler.inverse_transform(y_true, is_tensor=is_tensor, device=device)
        y_predicted = standard_scaler.inverse_transform(y_predicted, is_tensor=is_tensor, device=device)
    if loss_fn == "mae":
        return masked_mae_loss(y_predicted, y_true, mask_val=mask_val)
    return masked_mse_loss(y_predicted, y_true, mask_val=mask_val)


def cosine_annealing_lr(base_lr, epoch, num_epochs, eta_min=0.0):
    """Closed form of torch.optim.lr_scheduler.CosineAnnealingLR(T_max=num_epochs) stepped once per
    epoch (train.py:224,329; train_ssl.py:151,233): lr after `epoch` scheduler steps."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * epoch / num_epochs)) / 2.0


# ---- checkpoints ----------------------------------------------------------------------------------
def load_model_checkpoint(checkpoint_file, model, optimizer=None):
    """reference utils.py:156-163: restore `model_state` (and `optimizer_state` if asked)."""
    # the reference's checkpoints (utils.py:84-150) hold plain tensors / numbers only: no unpickling of arbitrary objects
    ckpt = torch.load(checkpoint_file, map_location="cpu", weights_only=True)
    model.load_state_dict(ckpt["model_state"])
    if optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer_state"])
        return model, optimizer
    return model


def build_finetune_model(model_new, model_pretrained, num_rnn_layers, num_layers_frozen=0):
    """reference utils.py:166-176: transplant the encoder dconv modules of the first
    `num_rnn_layers` cells from an SSL-pretrained model."""
    for layer in range(num_rnn_layers):
        src = model_pretrained.encoder.encoding_cells[layer]
        dst = model_new.encoder.encoding_cells[layer]
        dst.dconv_gate = src.dconv_gate
        dst.dconv_candidate = src.dconv_candidate
    return model_new


class CheckpointSaver:
    """`last.pth.tar` after every call, `best.pth.tar` whenever the tracked metric does not get worse
    (ties count as better): the file protocol of the reference (utils.py:84-150), so its scripts and
    `load_model_checkpoint` read these files unchanged.  `optimizer` is anything with a `state_dict()`
    (e.g. `TrainStep`)."""

    def __init__(self, save_dir, metric_name, maximize_metric=False, log=None):
        self.save_dir, self.metric_name, self.maximize_metric, self.log = save_dir, metric_name, maximize_metric, log
        self.best_val = None

    def is_best(self, metric_val):
        if metric_val is None:
            return False
        if self.best_val is None:
            return True
        return self.best_val <= metric_val if self.maximize_metric else self.best_val >= metric_val

    def save(self, epoch, model, optimizer, metric_val):
        import os
        import shutil
        last = os.path.join(self.save_dir, "last.pth.tar")
        torch.save({"epoch": epoch, "model_state": model.state_dict(), "optimizer_state": optimizer.state_dict()}, last)
        if self.is_best(metric_val):
            self.best_val = metric_val
            shutil.copy(last, os.path.join(self.save_dir, "best.pth.tar"))
            if self.log is not None:
                self.log.info(f"new best checkpoint at epoch {epoch} ({self.metric_name} = {metric_val})")


def eval_dict(y_pred, y, y_prob=None, file_names=None, average="macro"):
    """Score dictionary of the reference's evaluation (utils.py:285-317): accuracy, F1 / precision / recall
    with the given averaging, AUROC for binary problems when probabilities are given; plus the per-file
    prediction / label dictionaries."""
    from sklearn import metrics
    pred = dict(zip(file_names, y_pred)) if file_names is not None else {}
    true = dict(zip(file_names, y)) if file_names is not None else {}
    scores = {}
    if y is not None:
        scores["acc"] = metrics.accuracy_score(y, y_pred)
        scores["F1"] = metrics.f1_score(y, y_pred, average=average)
        scores["precision"] = metrics.precision_score(y, y_pred, average=average)
        scores["recall"] = metrics.recall_score(y, y_pred, average=average)
        if y_prob is not None and len(set(np.asarray(y).tolist())) <= 2:
            scores["auroc"] = metrics.roc_auc_score(y, y_prob)
    return scores, pred, true


def thresh_max_f1(y_true, y_prob):
    """Decision threshold that maximises F1 along the precision-recall curve (binary only;
    utils.py:320-343: the dev-set threshold search of the detection task)."""
    from sklearn.metrics import precision_recall_curve
    if len(set(np.asarray(y_true).tolist())) > 2:
        raise NotImplementedError("thresh_max_f1 is defined for binary labels")
    precision, recall, thresholds = precision_recall_curve(y_true, y_prob)
    p, r = precision[:len(thresholds)], recall[:len(thresholds)]
    with np.errstate(divide="ignore", invalid="ignore"):
        f1 = 2 * p * r / (p + r)
    keep = ~np.isnan(f1)
    return thresholds[keep][int(np.argmax(f1[keep]))]
