#!/usr/bin/env python3
"""Golden vectors for the TRAINING-mode dropout branches of the path (model/model.py:191 and :267; README.md:83 trains the
4-class model with --dropout 0.5), produced by running the GENUINE reference here:  python tests/golden/make_golden_dropout.py

nn.Dropout's own random draw cannot be reproduced by another implementation, so the reference models are run in train() mode
with their `dropout` submodule swapped for a module that multiplies by a CLOSED-FORM keep-mask x 1/(1-p) (closed_form.
cf_dropout_mask; a fresh one per call, like nn.Dropout) -- everything else, in particular WHERE the dropout sits
(fc(relu(dropout(last_out))); projection_layer(dropout(output)) at every decoder step), is the reference's own code.
Output: golden_dropout_v1.npz (logits / predictions, loss, parameter gradients).  Only executed, nothing copied."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
from closed_form import cf, cf_adjacency, cf_dropout_mask, cf_params, sample_view  # noqa: E402

for _m in ("h5py", "pyedflib"):
    sys.modules[_m] = types.ModuleType(_m)
sys.path.insert(0, REF)
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
from model.model import DCRNNModel_classification, DCRNNModel_nextTimePred  # noqa: E402
import utils as ref_utils  # noqa: E402
from data.data_utils import keep_topk  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(4)
N, P_DROP = 19, 0.5
G = {}
ADJ = np.load(os.path.join(HERE, "adj_mx_3d.npy"))
LAP = ref_utils.calculate_scaled_laplacian(ADJ, lambda_max=None).toarray()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class FixedMask(torch.nn.Module):
    """stands where nn.Dropout(p) stood: call k multiplies by the k-th closed-form mask"""

    def __init__(self, shape, phase0):
        super().__init__()
        self.shape, self.phase0, self.calls = shape, phase0, 0

    def forward(self, x):
        assert self.training
        m = T(cf_dropout_mask(self.shape, P_DROP, self.phase0 + 0.37 * self.calls))
        self.calls += 1
        return x * m.reshape(x.shape)


def make_args(**kw):
    d = dict(num_nodes=N, num_rnn_layers=2, rnn_units=64, input_dim=100, output_dim=100, max_diffusion_step=2,
             dcgru_activation="tanh", filter_type="laplacian", dropout=P_DROP, cl_decay_steps=3000, use_curriculum_learning=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def load_params(module, params):
    module.load_state_dict({k: T(v) for k, v in params.items()})


def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def lap_supports(b):
    return [torch.FloatTensor(LAP).unsqueeze(0).repeat(b, 1, 1)]


def dual_supports(b, phase0=0.3):
    s1, s2 = [], []
    for i in range(b):
        a = keep_topk(cf_adjacency(N, phase=phase0 + 1.7 * i), top_k=3, directed=True)
        s1.append(torch.FloatTensor(ref_utils.calculate_random_walk_matrix(a).T.toarray()))
        s2.append(torch.FloatTensor(ref_utils.calculate_random_walk_matrix(a.T).T.toarray()))
    return [torch.stack(s1), torch.stack(s2)]


def cls_case(tag, filt, din, h, classes, b, t, lengths, full=True):
    model = DCRNNModel_classification(make_args(filter_type=filt, input_dim=din, rnn_units=h), classes, device=None)
    assert isinstance(model.dropout, torch.nn.Dropout) and model.dropout.p == P_DROP
    load_params(model, cf_params(shapes_of(model), base_phase=2.3))
    model.dropout = FixedMask((b, N, h), phase0=0.9)
    sup = dual_supports(b) if filt == "dual_random_walk" else lap_supports(b)
    x = T(cf((b, t, N, din), scale=1.0, freq=0.4177, phase=0.9))
    for i, ln in enumerate(lengths):
        x[i, ln:] = 0
    model.train()
    logits = model(x, torch.LongTensor(lengths), sup)
    assert model.dropout.calls == 1
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


def ssl_case(tag, filt, din, h, layers, b, t_in, t_out, full=True):
    model = DCRNNModel_nextTimePred(make_args(filter_type=filt, input_dim=din, output_dim=din, rnn_units=h, num_rnn_layers=layers), device=None)
    assert isinstance(model.decoder.dropout, torch.nn.Dropout) and model.decoder.dropout.p == P_DROP
    params = cf_params(shapes_of(model), base_phase=3.7)
    for l in range(2, layers):
        for k in list(params):
            if k.startswith(f"decoder.decoding_cells.{l}."):
                params[k] = params[k.replace(f"decoding_cells.{l}.", "decoding_cells.1.")]
    load_params(model, params)
    model.decoder.dropout = FixedMask((b, N, h), phase0=1.3)
    sup = dual_supports(b) if filt == "dual_random_walk" else lap_supports(b)
    x = T(cf((b, t_in, N, din), scale=1.0, freq=0.4177, phase=0.9))
    y = T(cf((b, t_out, N, din), scale=1.0, freq=0.3319, phase=1.9))
    y[0, 0, 0, :3] = 0.0
    scaler = ref_utils.StandardScaler(mean=np.float64(3.924), std=np.float64(1.560))
    model.train()
    pred = model(x, y, sup, batches_seen=7)
    assert model.decoder.dropout.calls == t_out
    loss = ref_utils.compute_regression_loss(y_true=y, y_predicted=pred, loss_fn="MAE", standard_scaler=scaler, device=None)
    loss.backward()
    G[f"ssl/{tag}/loss"] = np.array(loss.item())
    G[f"ssl/{tag}/pred"] = pred.detach().numpy() if full else sample_view(pred.detach().numpy(), 7)
    for k, p in model.named_parameters():
        G[f"ssl/{tag}/d_{k}"] = p.grad.numpy().copy() if full else sample_view(p.grad.numpy())


cls_case("lap_small_ce_varlen", "laplacian", 8, 16, 4, 4, 6, [6, 3, 5, 1])
cls_case("dual_small_bce", "dual_random_walk", 8, 16, 1, 3, 5, [5, 5, 5])
cls_case("lap_default_ce_varlen", "laplacian", 100, 64, 4, 3, 8, [8, 5, 2], full=False)      # README.md:83's model
ssl_case("lap_small", "laplacian", 8, 16, 2, 3, 4, 3)
ssl_case("dual_small_L3", "dual_random_walk", 8, 16, 3, 2, 4, 3)
ssl_case("dual_default", "dual_random_walk", 100, 64, 2, 2, 5, 3, full=False)

np.savez_compressed(os.path.join(HERE, "golden_dropout_v1.npz"), **G)
print(f"wrote {len(G)} arrays ->", os.path.getsize(os.path.join(HERE, "golden_dropout_v1.npz")) / 1e3, "kB on disk")
