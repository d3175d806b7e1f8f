#!/usr/bin/env python3
"""Training-trajectory golden: the GENUINE reference model (imported from
/root/reference, build container only) trained for a few steps with the reference's step recipe
(train.py:222-224,253-275: Adam(lr, weight_decay) / BCEWithLogitsLoss / clip_grad_norm_(5.0)) on the
closed-form synthetic detection task of closed_form.train_task.  Stored: the loss of every step, the
final probabilities, AUROC, and fingerprints of two trained parameters -> golden_train_v1.npz.

Run once, here:  python tests/golden/make_golden_train.py"""
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
from closed_form import cf, cf_adjacency, cf_params, sample_view, train_task  # noqa: E402

for _m in ("h5py", "pyedflib"):
    sys.modules[_m] = types.ModuleType(_m)
sys.path.insert(0, REF)
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
from model.model import DCRNNModel_classification, DCRNNModel_nextTimePred  # noqa: E402
from data.data_utils import keep_topk  # noqa: E402
import utils as ref_utils  # noqa: E402
from sklearn.metrics import roc_auc_score  # noqa: E402

torch.set_num_threads(4)
STEPS, LR, WD, CLIP = 20, 1e-3, 5e-4, 1.0
B, T = 32, 12

with open(os.path.join(REF, "data/electrode_graph/adj_mx_3d.pkl"), "rb") as f:
    adj = pickle.load(f)[-1].astype(np.float32)
lap = torch.FloatTensor(ref_utils.calculate_scaled_laplacian(adj, lambda_max=None).toarray())
sup = [lap.unsqueeze(0).repeat(B, 1, 1)]

args = types.SimpleNamespace(num_nodes=19, num_rnn_layers=2, rnn_units=64, input_dim=100, output_dim=100,
                             max_diffusion_step=2, dcgru_activation="tanh", filter_type="laplacian", dropout=0.0,
                             cl_decay_steps=3000, use_curriculum_learning=False)
model = DCRNNModel_classification(args, 1, device=None)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in cf_params(shapes, base_phase=4.1).items()})
model.train()
x, y = (torch.from_numpy(a) for a in train_task(B, T))
lengths = torch.full((B,), T, dtype=torch.long)
opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=WD)
loss_fn = torch.nn.BCEWithLogitsLoss()
losses, norms = [], []
for _ in range(STEPS):
    opt.zero_grad()
    logits = model(x, lengths, sup)
    loss = loss_fn(logits.view(-1), y)
    loss.backward()
    norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), CLIP)))
    opt.step()
    losses.append(loss.item())
with torch.no_grad():
    prob = torch.sigmoid(model(x, lengths, sup)).view(-1).numpy()
G = {"train/losses": np.array(losses), "train/grad_norms": np.array(norms), "train/final_prob": prob,
     "train/auroc": np.array(roc_auc_score(y.numpy(), prob)),
     "train/hparams": np.array([STEPS, LR, WD, CLIP, B, T], dtype=np.float64)}
sd = model.state_dict()
for k in ("fc.weight", "encoder.encoding_cells.0.dconv_gate.weight", "encoder.encoding_cells.1.dconv_candidate.biases"):
    G[f"train/final/{k}"] = sample_view(sd[k].numpy(), 53)

# ---- SSL (encoder + autoregressive decoder, shared decoder cell, dual random walk, train_ssl.py:158-176) ----
S_STEPS, S_B, S_TIN, S_TOUT, S_LR = 12, 6, 6, 4, 5e-3
sargs = types.SimpleNamespace(num_nodes=19, num_rnn_layers=3, rnn_units=32, input_dim=20, output_dim=20,
                              max_diffusion_step=2, dcgru_activation="tanh", filter_type="dual_random_walk",
                              dropout=0.0, cl_decay_steps=3000, use_curriculum_learning=False)
ssl = DCRNNModel_nextTimePred(sargs, device=None)
shapes = {k: tuple(v.shape) for k, v in ssl.state_dict().items()}
raw = cf_params(shapes, base_phase=5.3)
for k in list(raw):                     # layers >= 1 of the decoder are ONE cell object (model.py:126-143)
    if k.startswith("decoder.decoding_cells.2."):
        raw[k] = raw[k.replace("decoding_cells.2.", "decoding_cells.1.")]
ssl.load_state_dict({k: torch.from_numpy(v) for k, v in raw.items()})
ssl.train()
s1, s2 = [], []
for i in range(S_B):
    a = keep_topk(cf_adjacency(19, phase=0.3 + 1.7 * i), top_k=3, directed=True)
    s1.append(torch.FloatTensor(ref_utils.calculate_random_walk_matrix(a).T.toarray()))
    s2.append(torch.FloatTensor(ref_utils.calculate_random_walk_matrix(a.T).T.toarray()))
ssup = [torch.stack(s1), torch.stack(s2)]
MEAN, STD = 3.924, 1.560
sx = torch.from_numpy(cf((S_B, S_TIN, 19, 20), scale=1.0, freq=0.4177, phase=0.9))
sy = torch.from_numpy(cf((S_B, S_TOUT, 19, 20), scale=1.0, freq=0.3319, phase=1.9))
sy[0, 0, 0, :3] = -MEAN / STD           # inverse-transforms to exactly 0 -> masked out (mask_val = 0)
scaler = ref_utils.StandardScaler(mean=np.float64(MEAN), std=np.float64(STD))
opt = torch.optim.Adam(ssl.parameters(), lr=S_LR, weight_decay=WD)
slosses, snorms = [], []
for it in range(S_STEPS):
    opt.zero_grad()
    pred = ssl(sx, sy, ssup, batches_seen=it * S_B)
    loss = ref_utils.compute_regression_loss(y_true=sy, y_predicted=pred, loss_fn="MAE", standard_scaler=scaler, device=None)
    loss.backward()
    snorms.append(float(torch.nn.utils.clip_grad_norm_(ssl.parameters(), CLIP)))
    opt.step()
    slosses.append(loss.item())
with torch.no_grad():
    G["ssl_train/final_pred"] = sample_view(ssl(sx, sy, ssup, batches_seen=0).numpy(), 31)
G["ssl_train/losses"], G["ssl_train/grad_norms"] = np.array(slosses), np.array(snorms)
G["ssl_train/hparams"] = np.array([S_STEPS, S_LR, WD, CLIP, S_B, S_TIN, S_TOUT, MEAN, STD], dtype=np.float64)
print("ssl losses", [round(v, 5) for v in slosses])
print("ssl norms", [round(v, 4) for v in snorms])
np.savez_compressed(os.path.join(HERE, "golden_train_v1.npz"), **G)
print("losses", [round(v, 5) for v in losses])
print("norms", [round(v, 4) for v in norms])
print("auroc", G["train/auroc"], "labels", int(y.sum()), "of", B)
