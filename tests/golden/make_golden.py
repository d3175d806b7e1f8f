#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the GENUINE reference
(tsy935/eeg-gnn-ssl at /root/reference) in the build container.

Run once, here (the GPU box has no /root/reference):  python tests/golden/make_golden.py
Outputs: golden_v1.npz (reference outputs / gradients), adj_mx_3d.npy (the reference's
19x19 electrode adjacency DATA asset), pretrained_manifest.json (state_dict key/shape
manifest of the four shipped checkpoints).  All inputs and weights are closed-form fills
(closed_form.py) so only outputs are stored.  Nothing from the reference's source text is
copied; the reference is only *executed*."""
import json
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
from closed_form import cf, cf_adjacency, cf_params, sample_view  # noqa: E402

for _m in ("h5py", "pyedflib"):          # imported by utils.py / data_utils.py, unused on this path
    sys.modules[_m] = types.ModuleType(_m)
sys.path.insert(0, REF)
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self   # model.py:336 hard-codes .cuda()
from model.cell import DCGRUCell, DiffusionGraphConv  # noqa: E402
from model.model import DCRNNModel_classification, DCRNNModel_nextTimePred  # noqa: E402
import utils as ref_utils  # noqa: E402
from data.data_utils import keep_topk, comp_xcorr  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(4)
N = 19
G = {}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_args(**kw):
    d = dict(num_nodes=N, num_rnn_layers=2, rnn_units=64, input_dim=100, output_dim=100,
             max_diffusion_step=2, dcgru_activation="tanh", filter_type="laplacian", dropout=0.0,
             cl_decay_steps=3000, use_curriculum_learning=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def load_params(module, params):
    sd = module.state_dict()
    assert set(sd.keys()) == set(params.keys()), (sorted(sd.keys()), sorted(params.keys()))
    module.load_state_dict({k: T(v) for k, v in params.items()})


def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


# ---- supports -----------------------------------------------------------------------------
with open(os.path.join(REF, "data/electrode_graph/adj_mx_3d.pkl"), "rb") as f:
    ADJ = pickle.load(f)[-1].astype(np.float32)
np.save(os.path.join(HERE, "adj_mx_3d.npy"), ADJ)
LAP = ref_utils.calculate_scaled_laplacian(ADJ, lambda_max=None).toarray()
G["supports/scaled_laplacian_adj3d"] = LAP.astype(np.float64)
G["supports/scaled_laplacian_adj3d_lmax2"] = ref_utils.calculate_scaled_laplacian(ADJ).toarray()


def lap_supports(b, batched=True):
    s = torch.FloatTensor(LAP)
    return [s.unsqueeze(0).repeat(b, 1, 1)] if batched else [s]


def dual_supports(b, phase0=0.3):
    s1, s2 = [], []
    for i in range(b):
        a = cf_adjacency(N, phase=phase0 + 1.7 * i)
        a = keep_topk(a, top_k=3, directed=True)
        s1.append(torch.FloatTensor(ref_utils.calculate_random_walk_matrix(a).T.toarray()))
        s2.append(torch.FloatTensor(ref_utils.calculate_random_walk_matrix(a.T).T.toarray()))
    return [torch.stack(s1), torch.stack(s2)]


# per-clip correlation graph pipeline on a closed-form clip (dataloader_detection.py:258-307,346-349)
clip = cf((12, N, 100), scale=1.0, freq=0.7391, phase=0.2) + cf((12, N, 100), scale=0.5, freq=0.0137, phase=1.0)
flat = np.transpose(clip, (1, 0, 2)).reshape(N, -1)
adj = np.eye(N, dtype=np.float32)
for i in range(N):
    for j in range(i + 1, N):
        xc = comp_xcorr(flat[i], flat[j], mode="valid", normalize=True)
        adj[i, j] = xc
        adj[j, i] = xc
adj = keep_topk(abs(adj), top_k=3, directed=True)
G["corr/adj"] = adj
G["corr/s1"] = ref_utils.calculate_random_walk_matrix(adj).T.toarray()
G["corr/s2"] = ref_utils.calculate_random_walk_matrix(adj.T).T.toarray()


# ---- DiffusionGraphConv ---------------------------------------------------------------------
def dconv_case(tag, filt, din, h, o, b, batched=True):
    ns = 2 if filt == "dual_random_walk" else 1
    mod = DiffusionGraphConv(num_supports=ns, input_dim=din, hid_dim=h, num_nodes=N,
                             max_diffusion_step=2, output_dim=o, filter_type=filt)
    load_params(mod, cf_params(shapes_of(mod), base_phase=0.5))
    sup = dual_supports(b) if ns == 2 else lap_supports(b, batched)
    x = T(cf((b, N * din), scale=1.0, freq=0.371, phase=0.1))
    s = T(cf((b, N * h), scale=0.8, freq=0.533, phase=0.7))
    with torch.no_grad():
        G[f"dconv/{tag}/out"] = mod(sup, x, s, o).numpy()


dconv_case("lap_small", "laplacian", 8, 16, 32, 3)
dconv_case("lap_small_unbatched", "laplacian", 8, 16, 32, 3, batched=False)
dconv_case("dual_small", "dual_random_walk", 8, 16, 32, 3)
dconv_case("lap_default", "laplacian", 100, 64, 128, 2)
dconv_case("dual_default", "dual_random_walk", 100, 64, 128, 2)


# ---- DCGRUCell fwd + grads --------------------------------------------------------------------
def cell_case(tag, filt, din, h, b, act="tanh", full=True):
    cell = DCGRUCell(input_dim=din, num_units=h, max_diffusion_step=2, num_nodes=N,
                     filter_type=filt, nonlinearity=act)
    load_params(cell, cf_params(shapes_of(cell), base_phase=1.1))
    sup = dual_supports(b) if filt == "dual_random_walk" else lap_supports(b)
    x = T(cf((b, N * din), scale=1.0, freq=0.371, phase=0.1)).requires_grad_(True)
    s = T(cf((b, N * h), scale=0.8, freq=0.533, phase=0.7)).requires_grad_(True)
    out, new = cell(sup, x, s)
    wout = T(cf((b, N * h), scale=1.0, freq=0.291, phase=0.4))
    (out * wout).sum().backward()
    G[f"cell/{tag}/out"] = out.detach().numpy()
    grads = {"dx": x.grad, "dh": s.grad}
    for k, p in cell.named_parameters():
        grads["d_" + k] = p.grad
    for k, g in grads.items():
        G[f"cell/{tag}/{k}"] = g.numpy() if full else sample_view(g.numpy())


cell_case("lap_small", "laplacian", 8, 16, 3)
cell_case("dual_small", "dual_random_walk", 8, 16, 3)
cell_case("lap_small_relu", "laplacian", 8, 16, 3, act="relu")
cell_case("lap_default", "laplacian", 100, 64, 2, full=False)
cell_case("dual_default", "dual_random_walk", 100, 64, 2, full=False)
cell_case("lap_l1_default", "laplacian", 64, 64, 2, full=False)


# ---- classification / detection model -----------------------------------------------------------
def cls_case(tag, filt, din, h, layers, classes, b, t, lengths=None, full=True):
    args = make_args(filter_type=filt, input_dim=din, rnn_units=h, num_rnn_layers=layers)
    model = DCRNNModel_classification(args, classes, device=None)
    load_params(model, cf_params(shapes_of(model), base_phase=2.3))
    sup = dual_supports(b) if filt == "dual_random_walk" else lap_supports(b)
    x = T(cf((b, t, N, din), scale=1.0, freq=0.4177, phase=0.9))
    if lengths is None:
        lengths = [t] * b
    else:  # zero padding after len, as dataloader_classification.py:333-343
        for i, ln in enumerate(lengths):
            x[i, ln:] = 0
    seq = torch.LongTensor(lengths)
    # expose the encoder outputs too
    model.train()
    logits = model(x, seq, sup)
    G[f"cls/{tag}/logits"] = logits.detach().numpy()
    if classes == 1:
        y = T((cf((b,), scale=1.0, freq=2.1, phase=0.3) > 0).astype(np.float32))
        loss = torch.nn.BCEWithLogitsLoss()(logits.view(-1), y)
    else:
        y = torch.LongTensor([(3 * i + 1) % classes for i in range(b)])
        loss = torch.nn.CrossEntropyLoss()(logits, y)
    loss.backward()
    G[f"cls/{tag}/loss"] = np.array(loss.item())
    for k, p in model.named_parameters():
        G[f"cls/{tag}/d_{k}"] = p.grad.numpy() if full else sample_view(p.grad.numpy())
    with torch.no_grad():
        h0 = model.encoder.init_hidden(b)
        fin, top = model.encoder(x.transpose(0, 1), h0, sup)
    G[f"cls/{tag}/enc_final"] = fin.numpy()
    G[f"cls/{tag}/enc_top"] = top.numpy() if (full or top.numel() < 12000) else sample_view(top.numpy(), 7)


cls_case("lap_small_bce", "laplacian", 8, 16, 2, 1, 3, 5)
cls_case("lap_small_ce_varlen", "laplacian", 8, 16, 2, 4, 4, 6, lengths=[6, 3, 5, 1])
cls_case("dual_small_bce", "dual_random_walk", 8, 16, 2, 1, 3, 5)
cls_case("lap_default_bce", "laplacian", 100, 64, 2, 1, 4, 12, full=False)          # BASELINE cfg1 shape
cls_case("dual_default_bce", "dual_random_walk", 100, 64, 2, 1, 3, 6, full=False)
cls_case("lap_default_ce_varlen", "laplacian", 100, 64, 2, 4, 3, 8, lengths=[8, 5, 2], full=False)


# ---- SSL seq2seq model ------------------------------------------------------------------------
def ssl_case(tag, filt, din, h, layers, b, t_in, t_out, full=True):
    args = make_args(filter_type=filt, input_dim=din, output_dim=din, rnn_units=h, num_rnn_layers=layers)
    model = DCRNNModel_nextTimePred(args, device=None)
    sd_shapes = shapes_of(model)
    params = cf_params(sd_shapes, base_phase=3.7)
    for l in range(2, layers):      # Q6: shared decoder cell -> identical tensors under both keys
        for k in list(params):
            if k.startswith(f"decoder.decoding_cells.{l}."):
                params[k] = params[k.replace(f"decoding_cells.{l}.", "decoding_cells.1.")]
    load_params(model, params)
    sup = dual_supports(b) if filt == "dual_random_walk" else lap_supports(b)
    x = T(cf((b, t_in, N, din), scale=1.0, freq=0.4177, phase=0.9))
    y = T(cf((b, t_out, N, din), scale=1.0, freq=0.3319, phase=1.9))
    y[0, 0, 0, :3] = 0.0     # exercise the mask (y_true == 0)
    scaler = ref_utils.StandardScaler(mean=np.float64(3.924), std=np.float64(1.560))
    model.train()
    for loss_name in ("MAE", "mae"):       # "MAE" -> masked RMSE (Q9), "mae" -> masked MAE
        model.zero_grad()
        pred = model(x, y, sup, batches_seen=7)
        loss = ref_utils.compute_regression_loss(y_true=y, y_predicted=pred, loss_fn=loss_name,
                                                 standard_scaler=scaler, device=None)
        loss.backward()
        G[f"ssl/{tag}/{loss_name}/loss"] = np.array(loss.item())
        seen = set()
        for k, p in model.named_parameters():
            G[f"ssl/{tag}/{loss_name}/d_{k}"] = p.grad.numpy().copy() if full else sample_view(p.grad.numpy())
            seen.add(k)
    G[f"ssl/{tag}/pred"] = pred.detach().numpy() if (full or pred.numel() < 12000) else sample_view(pred.detach().numpy(), 7)
    G[f"ssl/{tag}/named_parameters"] = np.array(sorted(seen))
    G[f"ssl/{tag}/state_dict_keys"] = np.array(sorted(sd_shapes))


ssl_case("lap_small", "laplacian", 8, 16, 2, 3, 4, 3)
ssl_case("dual_small_L3", "dual_random_walk", 8, 16, 3, 2, 4, 3)
ssl_case("dual_default", "dual_random_walk", 100, 64, 2, 2, 5, 3, full=False)

# ---- pretrained checkpoint manifest (state_dict contract) -----------------------------------------
manifest = {}
for fn in sorted(os.listdir(os.path.join(REF, "pretrained"))):
    ck = torch.load(os.path.join(REF, "pretrained", fn), map_location="cpu", weights_only=False)
    manifest[fn] = {"top_keys": sorted(ck.keys()),
                    "model_state": {k: list(v.shape) for k, v in ck["model_state"].items()}}
with open(os.path.join(HERE, "pretrained_manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1, sort_keys=True)

np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **G)
tot = sum(v.nbytes for v in G.values())
print(f"wrote {len(G)} arrays, {tot/1e3:.1f} kB raw ->",
      os.path.getsize(os.path.join(HERE, "golden_v1.npz")) / 1e3, "kB on disk")
