"""Parity checks of the eeg_gnn_ssl_amd modules against the golden vectors of the genuine
reference (tests/golden) and against the oracle.  The same functions run
  * on the GPU through libeeg_dcrnn_hip.so (tests/test_gpu_parity.py, `-m gpu`), and
  * on the CPU through the emulator build of the same kernel sources (tests/test_emu_parity.py).

Tolerance: north_star demands 1e-4 (fp32) w.r.t. the reference forward; the kernels only
re-associate fp32 sums (measured ~1e-6), so the suite asserts the stricter TOL below."""
import types

import numpy as np
import torch

import cases
from closed_form import cf, sample_view
from oracle import dcrnn_oracle as orc

TOL = 2e-5          # asserted (relative to the tensor's max magnitude); north_star bar is 1e-4
NORTH_STAR_TOL = 1e-4


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def assert_close(a, b, what, tol=TOL):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: max err {e:.3e} > {tol:.1e}"
    assert tol <= NORTH_STAR_TOL


def assert_close_scaled(a, b, what, tol=TOL):
    """relative to max |b| even when that is < 1 (gradients)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(np.abs(b).max()), 1e-6)
    e = float(np.abs(a - b).max()) / scale
    assert e <= tol, f"{what}: max err {e:.3e} (rel. to max) > {tol:.1e}"


def assert_view(arr, gold_view, what, step=97, tol=TOL):
    v = sample_view(arr, step)
    assert v.shape == gold_view.shape, (what, v.shape, gold_view.shape)
    scale = max(1e-6, float(np.abs(gold_view[3:]).max()))
    e = float(np.abs(v[3:] - gold_view[3:]).max()) / scale
    assert e <= tol, f"{what}: sampled max err {e:.3e}"
    assert abs(v[2] - gold_view[2]) <= 1e-4 * max(1e-9, gold_view[2]), f"{what}: sum of squares differs"


def make_args(cfg):
    return types.SimpleNamespace(num_nodes=cfg.num_nodes, num_rnn_layers=cfg.num_rnn_layers, rnn_units=cfg.rnn_units,
                                 input_dim=cfg.input_dim, output_dim=cfg.output_dim,
                                 max_diffusion_step=cfg.max_diffusion_step, dcgru_activation=cfg.dcgru_activation,
                                 filter_type=cfg.filter_type, dropout=0.0, cl_decay_steps=cfg.cl_decay_steps,
                                 use_curriculum_learning=False)


def load(module, params, device):
    missing = module.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)
    module.to(device)
    return missing


# ------------------------------------------------------------------------------------------------
def check_cell_case(tag, golden, adj3d, device):
    from eeg_gnn_ssl_amd import DCGRUCell
    c = cases.cell_inputs(tag, adj3d)
    cell = DCGRUCell(c["din"], c["h"], c["k"], 19, filter_type=c["filt"], nonlinearity=c["act"])
    load(cell, c["params"], device)
    x = c["x"].to(device).requires_grad_(True)
    s = c["s"].to(device).requires_grad_(True)
    sup = [t.to(device) for t in c["sup"]]
    out, new = cell(sup, x, s)
    assert out.shape == (c["b"], 19 * c["h"]) and new.shape == out.shape
    (out * c["wout"].to(device)).sum().backward()
    assert_close(out.detach().cpu().numpy(), golden[f"cell/{tag}/out"], f"cell/{tag}/out")
    grads = {"dx": x.grad, "dh": s.grad}
    grads.update({"d_" + k: p.grad for k, p in cell.named_parameters()})
    for k, g in grads.items():
        ref = golden[f"cell/{tag}/{k}"]
        if c["full"]:
            assert_close_scaled(g.cpu().numpy(), ref, f"cell/{tag}/{k}")
        else:
            assert_view(g.cpu().numpy(), ref, f"cell/{tag}/{k}")


def check_cls_case(tag, golden, adj3d, device):
    from eeg_gnn_ssl_amd import DCRNNModel_classification
    c = cases.cls_inputs(tag, adj3d)
    model = DCRNNModel_classification(make_args(c["cfg"]), c["classes"], device=device)
    load(model, c["params"], device)
    model.train()
    sup = [t.to(device) for t in c["sup"]]
    x = c["x"].to(device)
    logits = model(x, c["seq"].to(device), sup)
    assert_close(logits.detach().cpu().numpy(), golden[f"cls/{tag}/logits"], f"cls/{tag}/logits")
    y = c["y"].to(device)
    loss = (torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), y) if c["classes"] == 1
            else torch.nn.functional.cross_entropy(logits, y))
    loss.backward()
    assert abs(loss.item() - float(golden[f"cls/{tag}/loss"])) < 1e-5
    for k, p in model.named_parameters():
        ref = golden[f"cls/{tag}/d_{k}"]
        assert p.grad is not None, k
        if c["full"]:
            assert_close_scaled(p.grad.cpu().numpy(), ref, f"cls/{tag}/d_{k}")
        else:
            assert_view(p.grad.cpu().numpy(), ref, f"cls/{tag}/d_{k}")
    # encoder outputs through the public DCRNNEncoder.forward signature
    with torch.no_grad():
        b = x.shape[0]
        h0 = model.encoder.init_hidden(b).to(device)
        fin, top = model.encoder(x.transpose(0, 1), h0, sup)
    assert_close(fin.cpu().numpy(), golden[f"cls/{tag}/enc_final"], f"cls/{tag}/enc_final")
    ref_top = golden[f"cls/{tag}/enc_top"]
    if ref_top.ndim == 3:
        assert_close(top.cpu().numpy(), ref_top, f"cls/{tag}/enc_top")
    else:
        assert_view(top.cpu().numpy(), ref_top, f"cls/{tag}/enc_top", step=7)


def check_ssl_case(tag, golden, adj3d, device):
    from eeg_gnn_ssl_amd import DCRNNModel_nextTimePred, utils
    c = cases.ssl_inputs(tag, adj3d)
    model = DCRNNModel_nextTimePred(make_args(c["cfg"]), device=device)
    assert sorted(model.state_dict().keys()) == list(golden[f"ssl/{tag}/state_dict_keys"])
    assert sorted(k for k, _ in model.named_parameters()) == list(golden[f"ssl/{tag}/named_parameters"])
    load(model, c["params"], device)
    model.train()
    sup = [t.to(device) for t in c["sup"]]
    x, y = c["x"].to(device), c["y"].to(device)
    scaler = utils.StandardScaler(mean=np.float64(cases.SSL_MEAN), std=np.float64(cases.SSL_STD))
    for loss_name in ("MAE", "mae"):
        model.zero_grad()
        pred = model(x, y, sup, batches_seen=7)
        loss = utils.compute_regression_loss(y_true=y, y_predicted=pred, loss_fn=loss_name,
                                             standard_scaler=scaler, device=None)
        loss.backward()
        assert abs(loss.item() - float(golden[f"ssl/{tag}/{loss_name}/loss"])) < 2e-5, loss_name
        for k, p in model.named_parameters():
            ref = golden[f"ssl/{tag}/{loss_name}/d_{k}"]
            if c["full"]:
                assert_close_scaled(p.grad.cpu().numpy(), ref, f"ssl/{tag}/{loss_name}/d_{k}", tol=5e-5)
            else:
                assert_view(p.grad.cpu().numpy(), ref, f"ssl/{tag}/{loss_name}/d_{k}", tol=5e-5)
    ref_pred = golden[f"ssl/{tag}/pred"]
    if ref_pred.ndim == 4:
        assert_close(pred.detach().cpu().numpy(), ref_pred, f"ssl/{tag}/pred")
    else:
        assert_view(pred.detach().cpu().numpy(), ref_pred, f"ssl/{tag}/pred", step=7)


def check_dconv_case(tag, golden, adj3d, device):
    from eeg_gnn_ssl_amd import DiffusionGraphConv
    c = cases.dconv_inputs(tag, adj3d)
    ns = 2 if c["filt"] == "dual_random_walk" else 1
    mod = DiffusionGraphConv(ns, c["din"], c["h"], 19, 2, c["o"], filter_type=c["filt"])
    load(mod, {"weight": c["weight"], "biases": c["biases"]}, device)
    x, s = c["x"].to(device).requires_grad_(True), c["s"].to(device).requires_grad_(True)
    out = mod([t.to(device) for t in c["sup"]], x, s, c["o"])
    assert_close(out.detach().cpu().numpy(), golden[f"dconv/{tag}/out"], f"dconv/{tag}/out")
    # the module is differentiable like the reference's (cell.py:66-118): gradients w.r.t. inputs, state, weight, biases
    up = cases.T(cf((c["b"], 19 * c["o"]), scale=1.0, freq=0.291, phase=0.4)).to(device)
    (out * up).sum().backward()
    for got, key in ((x.grad, "dx"), (s.grad, "ds"), (mod.weight.grad, "d_weight"), (mod.biases.grad, "d_biases")):
        assert_close_scaled(got.cpu().numpy(), golden[f"dconv/{tag}/{key}"], f"dconv/{tag}/{key}")


def check_vs_oracle_random(device, filt, din, h, layers, t_len, b, classes, adj3d, seed=0, lengths=None, act="tanh", k=2):
    """Random-input model-level parity vs the oracle (logits + all parameter gradients)."""
    from eeg_gnn_ssl_amd import DCRNNModel_classification
    g = torch.Generator().manual_seed(seed)
    cfg = orc.DCRNNConfig(filter_type=filt, input_dim=din, rnn_units=h, num_rnn_layers=layers, num_classes=classes,
                          dcgru_activation=act, max_diffusion_step=k)
    params = orc.init_params(cfg, "classification", seed=seed)
    for k in params:
        if k.endswith("biases"):
            params[k] = 0.1 * torch.randn(params[k].shape, generator=g)
    sup = cases.supports_for(filt, adj3d, b)
    x = torch.randn(b, t_len, 19, din, generator=g)
    seq = torch.tensor(lengths if lengths is not None else [t_len] * b, dtype=torch.int64)
    y = (torch.rand(b, generator=g) > 0.5).float() if classes == 1 else torch.randint(0, classes, (b,), generator=g)
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    lo = orc.classification_forward(po, cfg, x, seq, sup)
    (orc.bce_with_logits(lo, y) if classes == 1 else orc.cross_entropy(lo, y)).backward()
    model = DCRNNModel_classification(make_args(cfg), classes, device=device)
    load(model, params, device)
    lg = model(x.to(device), seq.to(device), [s.to(device) for s in sup])
    yd = y.to(device)
    (torch.nn.functional.binary_cross_entropy_with_logits(lg.view(-1), yd) if classes == 1
     else torch.nn.functional.cross_entropy(lg, yd)).backward()
    assert_close(lg.detach().cpu().numpy(), lo.detach().numpy(), "logits vs oracle")
    for k, p in model.named_parameters():
        assert_close_scaled(p.grad.cpu().numpy(), po[k].grad.numpy(), f"d_{k} vs oracle", tol=5e-5)
    return {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}


def shared_laplacian(n, adj3d, g):
    """one symmetric scaled Laplacian (N,N): the distance graph for the 19-electrode montage, else of a random undirected graph"""
    if n == 19:
        return cases.supports_for("laplacian", adj3d, 1)[0][0].clone()
    a = torch.rand(n, n, generator=g).numpy().astype(np.float32)
    np.fill_diagonal(a, 1.0)
    return orc.compute_supports(a, "laplacian")[0]


def check_spectral_form(device, adj3d, din=100, layers=2, t_len=3, b=4, classes=1, k=2, n=19, seed=0, lengths=None, act="tanh"):
    """The spectral form of the hoisted x-part (csrc/spec_common.h): a model fed the ONE symmetric support in its shared (N,N) form
    runs every layer in the eigenbasis of that support -- logits and all parameter gradients against the oracle (which diffuses hop
    by hop on the batched form, cell.py:83-93) and against the general path of the same library (batched form of the same graph)."""
    from eeg_gnn_ssl_amd import DCRNNModel_classification, ops
    g = torch.Generator().manual_seed(seed)
    cfg = orc.DCRNNConfig(filter_type="laplacian", input_dim=din, rnn_units=64, num_rnn_layers=layers, num_classes=classes,
                          max_diffusion_step=k, num_nodes=n, dcgru_activation=act)
    params = orc.init_params(cfg, "classification", seed=seed)
    for name in params:
        if name.endswith("biases"):
            params[name] = 0.1 * torch.randn(params[name].shape, generator=g)
    s2 = shared_laplacian(n, adj3d, g)
    supb = [s2.unsqueeze(0).repeat(b, 1, 1)]
    x = torch.randn(b, t_len, n, din, generator=g)
    seq = torch.tensor(lengths if lengths is not None else [t_len] * b, dtype=torch.int64)
    y = (torch.rand(b, generator=g) > 0.5).float() if classes == 1 else torch.randint(0, classes, (b,), generator=g)
    po = {name: v.clone().requires_grad_(True) for name, v in params.items()}
    lo = orc.classification_forward(po, cfg, x, seq, supb)
    (orc.bce_with_logits(lo, y) if classes == 1 else orc.cross_entropy(lo, y)).backward()
    model = DCRNNModel_classification(make_args(cfg), classes, device=device)
    load(model, params, device)
    xd, sd, yd = x.to(device), seq.to(device), y.to(device)
    loss_of = lambda lg: (torch.nn.functional.binary_cross_entropy_with_logits(lg.view(-1), yd) if classes == 1   # noqa: E731
                          else torch.nn.functional.cross_entropy(lg, yd))
    shared = [s2.to(device)]
    assert ops.shared_spectral_basis(shared, k) is not None, "the scaled Laplacian of an undirected graph is symmetric"
    before = ops.spectral_layer_calls
    lg = model(xd, sd, shared)
    assert ops.spectral_layer_calls == before + layers, "every layer of the encoder takes the spectral form"
    loss_of(lg).backward()
    assert_close(lg.detach().cpu().numpy(), lo.detach().numpy(), "spectral logits vs oracle")
    spec = {}
    for name, p in model.named_parameters():
        assert_close_scaled(p.grad.cpu().numpy(), po[name].grad.numpy(), f"spectral d_{name} vs oracle", tol=5e-5)
        spec[name] = p.grad.detach().clone()
    model.zero_grad(set_to_none=True)
    lg2 = model(xd, sd, [t.to(device) for t in supb])       # batched form: the general path
    assert ops.spectral_layer_calls == before + layers
    loss_of(lg2).backward()
    assert_close(lg.detach().cpu().numpy(), lg2.detach().cpu().numpy(), "spectral vs general logits")
    for name, p in model.named_parameters():
        assert_close_scaled(spec[name].cpu().numpy(), p.grad.cpu().numpy(), f"spectral vs general d_{name}", tol=5e-5)
    # the same model with the mode switched off takes the general path on the shared form too, bit-equal to ... itself
    prev = ops.set_spectral_mode(0)
    try:
        model.zero_grad(set_to_none=True)
        lg3 = model(xd, sd, shared)
        assert ops.spectral_layer_calls == before + layers
    finally:
        ops.set_spectral_mode(prev)
    assert_close(lg3.detach().cpu().numpy(), lo.detach().numpy(), "general path on the shared form vs oracle")


def check_spectral_basis(device, adj3d):
    """`eeg_dcrnn_spectral_basis`: U orthonormal, U diag(lam) U^T = S, the Chebyshev table T_m(lam); a non-symmetric support is
    reported through the residual and the Python layer keeps the general path for it; batched / several / per-clip supports too."""
    from eeg_gnn_ssl_amd import ops
    g = torch.Generator().manual_seed(5)
    for n in (19, 2, 7, 20, 32):
        s2 = shared_laplacian(n, adj3d, g).to(device)
        blk = torch.ops.eeg_dcrnn.spectral_basis(s2).cpu().double()
        u, tc, info = blk[:n * n].view(n, n), blk[n * n:n * n + 256].view(8, 32), blk[n * n + 256:]
        lam = tc[1, :n]
        assert (u.T @ u - torch.eye(n, dtype=torch.float64)).abs().max() < 5e-7
        assert (u @ torch.diag(lam) @ u.T - s2.cpu().double()).abs().max() < 1e-6
        ref = torch.linalg.eigvalsh(s2.cpu().double())
        assert (torch.sort(lam).values - ref).abs().max() < 5e-7
        assert (tc[0, :n] - 1).abs().max() == 0 and (tc[2, :n] - (2 * lam * lam - 1)).abs().max() < 5e-7
        assert (tc[3, :n] - (4 * lam ** 3 - 3 * lam)).abs().max() < 1e-6 and (n == 32 or tc[:, n:].abs().max() == 0)
        assert info[0] < 1e-6 and info[1] < 1e-12 and abs(info[2] - s2.abs().max().item()) < 1e-6
        assert ops.shared_spectral_basis([s2], 2) is not None
        assert ops.shared_spectral_basis([s2], 2) is ops.shared_spectral_basis([s2], 2), "cached per support tensor"
        assert ops.shared_spectral_basis([s2], 0) is None
    rw = cases.supports_for("random_walk", adj3d, 1)[0][0].to(device) if hasattr(cases, "supports_for") else None
    skew = shared_laplacian(19, adj3d, g).to(device)
    skew[3, 5] += 1e-3                                       # a support that is not symmetric: residual far above rounding
    assert torch.ops.eeg_dcrnn.spectral_basis(skew)[19 * 19 + 256].item() > 1e-4
    assert ops.shared_spectral_basis([skew], 2) is None
    if rw is not None and (rw - rw.T).abs().max() > 1e-4:
        assert ops.shared_spectral_basis([rw], 2) is None
    s2 = shared_laplacian(19, adj3d, g).to(device)
    assert ops.shared_spectral_basis([s2.unsqueeze(0).repeat(3, 1, 1)], 2) is None, "batched supports are per-clip by declaration"
    assert ops.shared_spectral_basis([s2, s2], 2) is None
    # batched copies of one graph collapse to the shared form (once per tensor); per-clip graphs do not
    same = s2.unsqueeze(0).repeat(4, 1, 1)
    col = ops.collapse_shared_supports([same])
    assert col[0].dim() == 2 and torch.equal(col[0], s2) and ops.collapse_shared_supports([same])[0] is col[0]
    diff = same.clone()
    diff[2, 0, 1] += 0.5
    assert ops.collapse_shared_supports([diff])[0] is diff
    assert ops.collapse_shared_supports([same, same])[0] is same and ops.collapse_shared_supports(None) is None
    prev = ops.set_spectral_mode(0)
    try:
        assert ops.collapse_shared_supports([same.clone()])[0].dim() == 3 and ops.shared_spectral_basis([s2], 2) is None
    finally:
        ops.set_spectral_mode(prev)


def check_shape_sweep(device, n, h, filt, k, din=8, layers=2, t_len=3, b=2, classes=4, seed=0):
    """Model-level parity vs the oracle away from the 19-electrode defaults: other node counts (second MFMA
    node tile empty / partial / full), hidden sizes and hop counts.  Random graph of n nodes."""
    from eeg_gnn_ssl_amd import DCRNNModel_classification
    g = torch.Generator().manual_seed(seed + 17 * n + h + k)
    cfg = orc.DCRNNConfig(num_nodes=n, filter_type=filt, input_dim=din, rnn_units=h, num_rnn_layers=layers,
                          num_classes=classes, max_diffusion_step=k)
    params = orc.init_params(cfg, "classification", seed=seed)
    for name in params:
        if name.endswith("biases"):
            params[name] = 0.1 * torch.randn(params[name].shape, generator=g)
    sups = []
    for i in range(b):                                       # one random directed graph per clip
        a = torch.rand(n, n, generator=g).numpy().astype(np.float32)
        np.fill_diagonal(a, 1.0)
        sups.append(orc.compute_supports(a, filt))
    sup = [torch.stack([s[j] for s in sups]) for j in range(len(sups[0]))]
    x = torch.randn(b, t_len, n, din, generator=g)
    seq = torch.tensor([t_len] + [max(1, t_len - 1)] * (b - 1), dtype=torch.int64)
    y = torch.randint(0, max(classes, 2), (b,), generator=g)
    if classes == 1:
        y = y.float()
    po = {name: v.clone().requires_grad_(True) for name, v in params.items()}
    lo = orc.classification_forward(po, cfg, x, seq, sup)
    (orc.bce_with_logits(lo, y) if classes == 1 else orc.cross_entropy(lo, y)).backward()
    model = DCRNNModel_classification(make_args(cfg), classes, device=device)
    load(model, params, device)
    lg = model(x.to(device), seq.to(device), [s.to(device) for s in sup])
    (torch.nn.functional.binary_cross_entropy_with_logits(lg.view(-1), y.to(device)) if classes == 1
     else torch.nn.functional.cross_entropy(lg, y.to(device))).backward()
    assert_close(lg.detach().cpu().numpy(), lo.detach().numpy(), f"logits n={n} h={h} {filt} k={k}")
    for name, p in model.named_parameters():
        assert_close_scaled(p.grad.cpu().numpy(), po[name].grad.numpy(), f"d_{name} n={n} h={h} {filt} k={k}", tol=5e-5)


def check_training_tail(device):
    """HIP loss kernels and the fused clip+Adam step vs their torch definitions
    (train.py:203-206,273-275: BCEWithLogits / CrossEntropy, clip_grad_norm_(5), Adam + coupled L2)."""
    from eeg_gnn_ssl_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(37, generator=g)
    y = (torch.rand(37, generator=g) > 0.5).float()
    xd = x.clone().to(device).requires_grad_(True)
    loss = ops.bce_with_logits(xd, y.to(device))
    loss.backward()
    xr = x.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(xr, y)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-6
    assert_close_scaled(xd.grad.cpu().numpy(), xr.grad.numpy(), "bce dlogits", tol=1e-5)
    x = torch.randn(21, 4, generator=g)
    yc = torch.randint(0, 4, (21,), generator=g)
    xd = x.clone().to(device).requires_grad_(True)
    loss = ops.cross_entropy(xd, yc.to(device))
    loss.backward()
    xr = x.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, yc)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-6
    assert_close_scaled(xd.grad.cpu().numpy(), xr.grad.numpy(), "ce dlogits", tol=1e-5)
    # masked MAE / RMSE (utils.py:431-495) with and without the scalar scaler; exact zeros of y_true are masked
    for shape, scaler in (((3, 4, 19, 10), (3.924, 1.56)), ((7001,), None), ((5, 12, 19, 100), (0.0, 1.0))):
        pr = torch.randn(shape, generator=g)
        yt = torch.randn(shape, generator=g)
        yt[torch.rand(shape, generator=g) < 0.2] = 0.0 if scaler is None else -scaler[0] / scaler[1]
        for name in ("mae", "MAE"):
            pd = pr.clone().to(device).requires_grad_(True)
            loss = ops.masked_regression_loss(pd, yt.to(device), None if scaler is None else scaler[0],
                                              None if scaler is None else scaler[1], name)
            loss.backward()
            po = pr.clone().requires_grad_(True)
            ref = orc.regression_loss(yt, po, None if scaler is None else scaler[0], None if scaler is None else scaler[1], name)
            ref.backward()
            assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item())), (shape, name, loss.item(), ref.item())
            assert_close_scaled(pd.grad.cpu().numpy(), po.grad.numpy(), f"masked {name} dpred", tol=1e-5)
    n = 70001
    p0 = torch.randn(n, generator=g)
    pd = p0.to(device).clone()
    m = torch.zeros(n, device=device)
    v = torch.zeros(n, device=device)
    ws = torch.zeros(64, device=device)
    norm = torch.zeros(1, device=device)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=3e-3, weight_decay=5e-4)
    for step in range(1, 5):
        gr = torch.randn(n, generator=g) * (0.001 if step == 3 else 0.05)      # step 3: no clipping
        gd = gr.to(device).clone()
        ops.clip_adam_step(pd, gd, m, v, step, 3e-3, (0.9, 0.999), 1e-8, 5e-4, 5.0, 1.0, ws, norm)
        pr.grad = gr.clone()
        nr = torch.nn.utils.clip_grad_norm_([pr], 5.0)
        opt.step()
        assert abs(norm.item() - nr.item()) <= 1e-4 * max(1.0, nr.item())
        assert_close_scaled(gd.cpu().numpy(), pr.grad.numpy(), f"clipped grad step {step}", tol=1e-5)
        assert (pd.cpu() - pr.detach()).abs().max().item() < 2e-6, step


def random_supports(n, b, filt, g):
    """one random directed graph of n nodes per clip -> the filter type's batched supports"""
    per_clip = []
    for _ in range(b):
        a = torch.rand(n, n, generator=g).numpy().astype(np.float32)
        np.fill_diagonal(a, 1.0)
        per_clip.append(orc.compute_supports(a, filt))
    return [torch.stack([s[j] for s in per_clip]) for j in range(len(per_clip[0]))]


def check_decoder_vs_oracle(device, filt, dout, h, layers, t_out, b, adj3d, seed=0, ratio=None, act="tanh", n=19, order=2):
    """DCGRUDecoder (the native decoder operator) vs the oracle on random inputs: outputs, gradient
    w.r.t. the initial hidden states and all parameter gradients (shared cell for layers >= 1),
    with the teacher-forcing coin flips (model.py:194-200) replayed from the same `random` seed."""
    import random
    from eeg_gnn_ssl_amd import DCGRUDecoder
    g = torch.Generator().manual_seed(seed)
    cfg = orc.DCRNNConfig(filter_type=filt, input_dim=dout, output_dim=dout, rnn_units=h, num_rnn_layers=layers,
                          dcgru_activation=act, num_nodes=n, max_diffusion_step=order)
    params = {k: v for k, v in orc.init_params(cfg, "ssl", seed=seed).items() if k.startswith("decoder.")}
    for k in params:
        if k.endswith("biases") and not any(params[k] is params[q] for q in params if q < k):
            params[k].copy_(0.1 * torch.randn(params[k].shape, generator=g))
    sup = cases.supports_for(filt, adj3d, b) if n == 19 else random_supports(n, b, filt, g)
    targets = torch.randn(t_out, b, n, dout, generator=g)
    h0 = 0.5 * torch.randn(layers, b, n * h, generator=g)
    wout = torch.randn(t_out, b, n * dout, generator=g)
    mask = None
    device_flags = ratio == "device"        # the flags as a DEVICE int32[T] tensor (read by the persistent kernels when they start)
    if device_flags:
        mask = [(3 * i + seed) % 5 in (0, 3) for i in range(t_out)]
        ratio = None
    elif ratio is not None:
        random.seed(seed)
        mask = [random.random() < ratio for _ in range(t_out)]
        assert any(mask) and not all(mask[:-1]), mask
    uniq = {}
    po = {}
    for k, v in params.items():                       # shared tensors stay shared in the autograd graph
        key = v.data_ptr()
        if key not in uniq:
            uniq[key] = v.clone().requires_grad_(True)
        po[k] = uniq[key]
    h0o = h0.clone().requires_grad_(True)
    oo = orc.decoder_forward(po, cfg, targets, h0o, sup, mask)
    (oo * wout).sum().backward()
    dec = DCGRUDecoder(input_dim=dout, max_diffusion_step=order, num_nodes=n, hid_dim=h, output_dim=dout,
                       num_rnn_layers=layers, dcgru_activation=act, filter_type=filt)
    load(dec, {k[len("decoder."):]: v for k, v in params.items()}, device)
    dec.train()
    h0d = h0.clone().to(device).requires_grad_(True)
    if ratio is not None:
        random.seed(seed)
    flags = torch.tensor([1 if v else 0 for v in mask], dtype=torch.int32, device=device) if device_flags else None
    out = dec(targets.to(device), h0d, [s.to(device) for s in sup], teacher_forcing_ratio=ratio, teacher_flags=flags)
    (out * wout.to(device)).sum().backward()
    assert_close(out.detach().cpu().numpy(), oo.detach().numpy(), "decoder outputs vs oracle")
    assert_close_scaled(h0d.grad.cpu().numpy(), h0o.grad.numpy(), "d_initial_hidden_state vs oracle", tol=5e-5)
    for k, p in dec.named_parameters():
        assert_close_scaled(p.grad.cpu().numpy(), po["decoder." + k].grad.numpy(), f"d_{k} vs oracle", tol=5e-5)


def check_correlation_supports(device, golden):
    """On-device per-clip correlation graph -> dual random-walk supports vs (i) the golden of the
    genuine reference pipeline on the closed-form clip and (ii) the oracle on random clips,
    including a silent electrode (zero norm), a short clip and ragged sizes."""
    from closed_form import cf
    from eeg_gnn_ssl_amd import ops
    clip = cf((12, 19, 100), scale=1.0, freq=0.7391, phase=0.2) + cf((12, 19, 100), scale=0.5, freq=0.0137, phase=1.0)
    x = torch.from_numpy(np.ascontiguousarray(clip, dtype=np.float32)).unsqueeze(0).to(device)
    (s1, s2), adj = ops.correlation_supports(x, top_k=3, return_adj=True)
    assert np.abs(adj[0].cpu().numpy() - golden["corr/adj"]).max() <= 2e-6
    assert np.abs(s1[0].cpu().numpy() - golden["corr/s1"]).max() <= 2e-6
    assert np.abs(s2[0].cpu().numpy() - golden["corr/s2"]).max() <= 2e-6
    g = torch.Generator().manual_seed(9)
    for (b, t_len, n, d, top_k) in ((5, 7, 19, 100, 3), (3, 60, 19, 100, 3), (2, 1, 19, 8, 2), (4, 9, 12, 20, 4),
                                    (2, 60, 32, 128, 31), (3, 2, 4, 4, 0), (300, 3, 19, 100, 3)):
        xs = torch.randn(b, t_len, n, d, generator=g)
        xs[0, :, 3, :] = 0.0                                  # a silent electrode: zero norm -> raw (zero) correlation
        xs[-1] = xs[-1] * 0.5 + xs[-1, :, :1, :]              # strongly correlated channels
        (s1, s2), adj = ops.correlation_supports(xs.to(device), top_k=top_k, return_adj=True)
        for i in range(b):
            a_ref = orc.correlation_adjacency(xs[i].numpy(), top_k=top_k)
            sup = [orc.random_walk(a_ref).T, orc.random_walk(a_ref.T).T]
            got = adj[i].cpu().numpy()
            # a top-k tie broken the other way would move whole entries; values elsewhere agree to fp32 rounding
            assert ((got != 0) == (a_ref != 0)).all(), (b, t_len, n, d, i)
            assert np.abs(got - a_ref).max() <= 5e-6
            assert np.abs(s1[i].cpu().numpy() - sup[0]).max() <= 5e-6
            assert np.abs(s2[i].cpu().numpy() - sup[1]).max() <= 5e-6


def check_grad_sink(device, adj3d):
    """TrainStep writes parameter gradients straight into its flat bucket (ops.GradSink): the bucket must
    equal the gradients of the ordinary autograd path, for the classification and the SSL model
    (shared decoder cell: two uses of one parameter set)."""
    from eeg_gnn_ssl_amd import DCRNNModel_classification, DCRNNModel_nextTimePred
    from eeg_gnn_ssl_amd.train_step import TrainStep
    g = torch.Generator().manual_seed(21)
    for task, layers in (("classification", 2), ("ssl", 3)):
        cfg = orc.DCRNNConfig(filter_type="dual_random_walk", input_dim=8, output_dim=8, rnn_units=16,
                              num_rnn_layers=layers, num_classes=4)
        torch.manual_seed(3)
        if task == "ssl":
            model = DCRNNModel_nextTimePred(make_args(cfg), device=device).to(device).train()
            y = torch.randn(3, 4, 19, 8, generator=g).to(device)
        else:
            model = DCRNNModel_classification(make_args(cfg), 4, device=device).to(device).train()
            y = torch.randint(0, 4, (3,), generator=g).to(device)
        x = torch.randn(3, 6, 19, 8, generator=g).to(device)
        lengths = torch.tensor([6, 4, 5]).to(device)
        sup = [s.to(device) for s in cases.supports_for("dual_random_walk", adj3d, 3)]
        ts = TrainStep(model, task=task)
        ts.forward_backward(x, y, lengths, sup)
        sunk = ts.fp.flat_grad.clone()
        ts.fp.zero_grad()
        out = model(x, y, sup) if task == "ssl" else model(x, lengths, sup)
        ts.loss(out, y).backward()                      # no sink: autograd accumulates into the views
        plain = ts.fp.flat_grad.clone()
        assert torch.equal(sunk, plain), (task, (sunk - plain).abs().max().item())


def check_plane_handover(device, adj3d):
    """Layers >= 1 take their input hop planes from the recurrent kernel of the layer below (slots 1..T of
    its Hplanes) instead of diffusing the hidden sequence again: same logits and gradients as with the
    hand-over switched off, and the hand-over must actually happen (training only: inference keeps no planes)."""
    from eeg_gnn_ssl_amd import DCRNNModel_classification, ops
    g = torch.Generator().manual_seed(5)
    for filt in ("dual_random_walk", "laplacian"):              # shared graph / one graph per clip
        cfg = orc.DCRNNConfig(filter_type=filt, input_dim=8, rnn_units=16, num_rnn_layers=3, num_classes=4)
        torch.manual_seed(2)
        model = DCRNNModel_classification(make_args(cfg), 4, device=device).to(device).train()
        x = torch.randn(3, 5, 19, 8, generator=g).to(device)
        lengths = torch.tensor([5, 2, 4]).to(device)
        sup = [s.to(device) for s in cases.supports_for(filt, adj3d, 3)]
        res = []
        real_layer = ops.dcgru_layer_ex

        def no_handover(*a, **kw):                      # test-side switch: drop the planes handed over by the layer below
            if len(a) >= 15:                            # (x_planes is the 15th positional parameter of dcgru_layer_ex)
                a = a[:14] + (None,) + a[15:]
            else:
                kw["x_planes"] = None
            return real_layer(*a, **kw)

        from eeg_gnn_ssl_amd.model import model as model_mod
        for on in (True, False):
            patched = [(m, m.dcgru_layer_ex) for m in (ops, model_mod) if hasattr(m, "dcgru_layer_ex")]
            if not on:
                for m, _ in patched:
                    m.dcgru_layer_ex = no_handover
            try:
                before = ops.hop_plane_handovers
                model.zero_grad()
                out = model(x, lengths, sup)
                out.square().sum().backward()
                assert ops.hop_plane_handovers - before == (2 if on else 0)
                res.append((out.detach().clone(), [q.grad.clone() for q in model.parameters()]))
            finally:
                for m, f in patched:
                    m.dcgru_layer_ex = f
        assert_close(res[0][0].cpu().numpy(), res[1][0].cpu().numpy(), f"{filt} logits", tol=1e-6)
        for a, b_, (nm, _) in zip(res[0][1], res[1][1], model.named_parameters()):
            assert_close_scaled(a.cpu().numpy(), b_.cpu().numpy(), f"{filt} grad {nm}", tol=1e-6)
        with torch.no_grad():                               # inference: nothing saved, nothing handed over
            before = ops.hop_plane_handovers
            model(x, lengths, sup)
            assert ops.hop_plane_handovers == before


def check_training_trajectory(device, golden_train, adj3d, steps=None):
    """TrainStep (HIP forward/backward + fused clip/Adam) follows the loss and gradient-norm trajectory of the
    GENUINE reference trained with its own recipe (tests/golden/make_golden_train.py), and ends at the same
    probabilities."""
    from closed_form import cf_params
    from eeg_gnn_ssl_amd import DCRNNModel_classification
    from eeg_gnn_ssl_amd.train_step import TrainStep
    c = cases.train_inputs(adj3d)
    n_steps, lr, wd, clip = (float(v) for v in golden_train["train/hparams"][:4])
    n_steps = int(n_steps) if steps is None else steps
    model = DCRNNModel_classification(make_args(c["cfg"]), 1, device=device)
    shapes = orc.param_shapes(c["cfg"], "classification")
    load(model, {k: torch.from_numpy(v) for k, v in cf_params(shapes, base_phase=c["base_phase"]).items()}, device)
    model.train()
    ts = TrainStep(model, task="detection", lr=lr, weight_decay=wd, max_grad_norm=clip)
    x, y, seq = c["x"].to(device), c["y"].to(device), c["seq"].to(device)
    sup = [s.to(device) for s in c["sup"]]
    for i in range(n_steps):
        loss = ts.step(x, y, seq, sup).item()
        tol = 2e-5 * (1 + i)                                # rounding differences compound through Adam
        assert abs(loss - golden_train["train/losses"][i]) <= tol, (i, loss, golden_train["train/losses"][i])
        assert abs(ts.grad_norm.item() - golden_train["train/grad_norms"][i]) <= 10 * tol, (i, ts.grad_norm.item())
    if n_steps == int(golden_train["train/hparams"][0]):
        with torch.no_grad():
            prob = torch.sigmoid(model(x, seq, sup)).view(-1).cpu().numpy()
        assert np.abs(prob - golden_train["train/final_prob"]).max() <= 2e-3
        from sklearn.metrics import roc_auc_score         # north_star: "parity AUROC on synthetic labels"
        assert abs(roc_auc_score(c["y"].numpy(), prob) - float(golden_train["train/auroc"])) <= 1e-3
        from closed_form import cf, sample_view
        for k, v in model.state_dict().items():
            key = f"train/final/{k}"
            if key in golden_train:
                got, ref = sample_view(v.cpu().numpy(), 53), golden_train[key]
                assert np.abs(got[3:] - ref[3:]).max() <= 2e-3 * np.abs(ref[3:]).max(), k


def check_ssl_training_trajectory(device, golden_train, steps=None):
    """TrainStep(task="ssl") (native decoder operator, masked-RMSE kernel, fused clip/Adam) follows the
    genuine reference's SSL training trajectory (train_ssl.py recipe, shared decoder cell)."""
    from closed_form import cf, sample_view
    from eeg_gnn_ssl_amd import DCRNNModel_nextTimePred
    from eeg_gnn_ssl_amd.train_step import TrainStep
    c = cases.ssl_train_inputs(golden_train)
    model = DCRNNModel_nextTimePred(make_args(c["cfg"]), device=device)
    load(model, c["params"], device)
    model.train()
    ts = TrainStep(model, task="ssl", lr=c["lr"], weight_decay=c["wd"], max_grad_norm=c["clip"],
                   scaler_mean=c["mean"], scaler_std=c["std"])
    x, y = c["x"].to(device), c["y"].to(device)
    sup = [s.to(device) for s in c["sup"]]
    n_steps = c["steps"] if steps is None else steps
    for i in range(n_steps):
        loss = ts.step(x, y, None, sup).item()
        tol = 2e-5 * (1 + i)
        assert abs(loss - golden_train["ssl_train/losses"][i]) <= tol, (i, loss, golden_train["ssl_train/losses"][i])
        assert abs(ts.grad_norm.item() - golden_train["ssl_train/grad_norms"][i]) <= 10 * tol, (i, ts.grad_norm.item())
    if n_steps == c["steps"]:
        with torch.no_grad():
            pred = model(x, y, sup).cpu().numpy()
        assert np.abs(sample_view(pred, 31)[3:] - golden_train["ssl_train/final_pred"][3:]).max() <= 2e-3


def check_eval_driver(device, adj3d):
    """train_step.evaluate / predict (the reference's evaluation pass, train.py:332-431) against the same
    quantities computed by hand from the oracle's logits: sample-weighted loss, thresholded / arg-max
    predictions, the dev-set threshold search, score dictionary in the reference's order."""
    from sklearn import metrics
    from eeg_gnn_ssl_amd import DCRNNModel_classification, utils
    from eeg_gnn_ssl_amd.train_step import evaluate, predict
    g = torch.Generator().manual_seed(12)
    for task, classes in (("detection", 1), ("classification", 4)):
        cfg = orc.DCRNNConfig(filter_type="laplacian", input_dim=8, rnn_units=16, num_rnn_layers=2, num_classes=classes)
        params = orc.init_params(cfg, "classification", seed=3)
        model = DCRNNModel_classification(make_args(cfg), classes, device=device)
        load(model, params, device)
        model.train()                                       # evaluate() must switch to eval mode and back
        batches, ref_logits, ref_y = [], [], []
        for b in (5, 3):                                    # two batches of different size: the loss is sample-weighted
            x = torch.randn(b, 4, 19, 8, generator=g)
            seq = torch.randint(1, 5, (b,), generator=g)
            y = (torch.rand(b, generator=g) > 0.5).float() if classes == 1 else torch.randint(0, classes, (b,), generator=g)
            sup = cases.supports_for("laplacian", adj3d, b)
            batches.append((x.to(device), y.to(device), seq.to(device), [s.to(device) for s in sup]))
            ref_logits.append(orc.classification_forward(params, cfg, x, seq, sup))
            ref_y.append(y)
        lo, yy = torch.cat(ref_logits), torch.cat(ref_y)
        res = evaluate(model, batches, task=task, is_test=True, eval_set="dev")
        assert model.training
        assert list(res.keys())[:6] == ["loss", "acc", "F1", "recall", "precision", "best_thresh"]
        if classes == 1:
            ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(lo.view(-1), yy).item()
            prob = torch.sigmoid(lo.view(-1)).numpy()
            thr = utils.thresh_max_f1(y_true=yy.numpy().astype(int), y_prob=prob)
            pred = (prob > thr).astype(int)
            assert abs(res["best_thresh"] - thr) < 1e-5
            assert abs(res["auroc"] - metrics.roc_auc_score(yy.numpy().astype(int), prob)) < 1e-6
            assert abs(res["F1"] - metrics.f1_score(yy.numpy().astype(int), pred, average="binary")) < 1e-6
        else:
            ref_loss = torch.nn.functional.cross_entropy(lo, yy).item()
            pred = lo.argmax(dim=1).numpy()
            assert "auroc" not in res and res["best_thresh"] == 0.5
            assert abs(res["F1"] - metrics.f1_score(yy.numpy(), pred, average="weighted")) < 1e-6
        assert abs(res["loss"] - ref_loss) < 1e-5, (task, res["loss"], ref_loss)
        assert abs(res["acc"] - metrics.accuracy_score(yy.numpy().astype(int), pred)) < 1e-6
        y_prob, y_true = predict(model, batches, task=task)
        ref_prob = torch.sigmoid(lo.view(-1)).numpy() if classes == 1 else torch.softmax(lo, dim=1).numpy()
        assert np.abs(y_prob - ref_prob).max() < 1e-5 and (y_true == yy.numpy()).all()


def check_ssl_eval_driver(device, adj3d):
    """train_step.evaluate_ssl (train_ssl.py:232-280): eval-mode predictions, masked MAE in original units per batch,
    batch-size-weighted average -- against the same quantities from the oracle."""
    from eeg_gnn_ssl_amd import DCRNNModel_nextTimePred
    from eeg_gnn_ssl_amd.train_step import evaluate_ssl
    g = torch.Generator().manual_seed(21)
    cfg = orc.DCRNNConfig(filter_type="dual_random_walk", input_dim=20, output_dim=20, rnn_units=64, num_rnn_layers=2)
    params = orc.init_params(cfg, "ssl", seed=4)
    a = make_args(cfg)
    a.dropout, a.use_curriculum_learning = 0.5, True      # both must be inert in eval mode
    model = DCRNNModel_nextTimePred(a, device=device)
    load(model, params, device)
    model.train()
    batches, num, den, ref_preds = [], 0.0, 0, []
    for b in (4, 2, 3):
        x = torch.randn(b, 5, 19, 20, generator=g)
        y = torch.randn(b, 3, 19, 20, generator=g)
        y[0, 1, 5, :7] = -cases.SSL_MEAN / cases.SSL_STD                              # entries that un-scale to 0 are masked out
        sup = cases.supports_for("dual_random_walk", adj3d, b)
        batches.append((x.to(device), y.to(device), [s.to(device) for s in sup]))
        pr = orc.next_time_pred_forward(params, cfg, x, y, sup)
        ref_preds.append(pr)
        num += orc.regression_loss(y, pr, cases.SSL_MEAN, cases.SSL_STD, loss_fn="mae").item() * b
        den += b
    loss, preds, truths = evaluate_ssl(model, batches, cases.SSL_MEAN, cases.SSL_STD, return_predictions=True)
    assert model.training
    assert abs(loss - num / den) < 1e-5 * max(1.0, abs(num / den)), (loss, num / den)
    assert preds.shape == (9, 3, 19, 20) and truths.shape == preds.shape
    assert_close(preds, torch.cat(ref_preds).detach().numpy(), "evaluate_ssl predictions")
    assert abs(evaluate_ssl(model, batches[:1], cases.SSL_MEAN, cases.SSL_STD) -
               orc.regression_loss(batches[0][1].cpu(), ref_preds[0], cases.SSL_MEAN, cases.SSL_STD, loss_fn="mae").item()) < 1e-5


def check_fft_features(device, golden_fft):
    """On-device featurisation (1-s windows -> log|FFT| -> reflection / amplitude jitter -> z-score) vs the
    goldens of the genuine reference pipeline and, for the augmented variant and a ragged shape, the oracle."""
    from closed_form import fft_raw_signal
    from eeg_gnn_ssl_amd import ops
    mean, std = (float(v) for v in golden_fft["fft/mean_std"])
    raw64 = fft_raw_signal()
    raw = torch.from_numpy(raw64.astype(np.float32)).unsqueeze(0).to(device)          # (1, 19, 800)
    feat_raw, feat_std = ops.fft_features(raw, window=200, mean=mean, std=std)
    assert feat_raw.shape == (1, 4, 19, 100)
    # the operator takes float32 signals (the reference float64): rounding the samples to float32 moves the
    # weakest bins (|X| ~ 1 next to 30-uV components) by ~1e-5 in the log; the transform itself is fp64
    assert np.abs(feat_raw[0].cpu().numpy() - golden_fft["fft/logamp"]).max() <= 3e-5
    assert np.abs(feat_std[0].cpu().numpy() - golden_fft["fft/standardized"]).max() <= 3e-5
    same_input = orc.fft_features(raw64.astype(np.float32).astype(np.float64), window=200)
    assert np.abs(feat_raw[0].cpu().numpy() - same_input).max() <= 2e-6           # same (rounded) samples: fp32 output rounding only
    assert abs(feat_raw[0, 1, 5, 0].item() - np.log(1e-8)) < 1e-5                      # silent window: amp == 0 -> 1e-8
    # augmentation: left/right reflection (node permutation) + amplitude jitter, two clips, other window length
    g = torch.Generator().manual_seed(2)
    rawb = torch.randn(2, 19, 3 * 40, generator=g) * 20.0
    perm = torch.stack([torch.arange(19), torch.arange(19)]).to(torch.int32)
    perm[1, [0, 1]] = torch.tensor([1, 0], dtype=torch.int32)                         # swap one channel pair in clip 1
    perm[1, [4, 7]] = torch.tensor([7, 4], dtype=torch.int32)
    ls = torch.tensor([0.0, float(np.log(1.137))])
    fr, fs = ops.fft_features(rawb.to(device), window=40, mean=0.5, std=2.0, perm=perm.to(device), log_scale=ls.to(device))
    for b in range(2):
        ref = orc.fft_features(rawb[b].numpy().astype(np.float64), window=40)          # (3, 19, 20)
        assert np.abs(fr[b].cpu().numpy() - ref).max() <= 5e-6
        exp = ((ref[:, perm[b].numpy(), :] + float(ls[b])) - 0.5) / 2.0
        assert np.abs(fs[b].cpu().numpy() - exp).max() <= 5e-6
    # the same augmentation through the 200-sample kernel (mixed-radix transform, six windows per wave: 2 x 19 x 5 = 190 windows
    # = 31 full groups + one of 4), feat_raw at the source channel's slot
    rawb = torch.randn(2, 19, 5 * 200, generator=g) * 20.0
    fr, fs = ops.fft_features(rawb.to(device), window=200, mean=0.5, std=2.0, perm=perm.to(device), log_scale=ls.to(device))
    for b in range(2):
        ref = orc.fft_features(rawb[b].numpy().astype(np.float64), window=200)         # (5, 19, 100)
        assert np.abs(fr[b].cpu().numpy() - ref).max() <= 5e-6
        exp = ((ref[:, perm[b].numpy(), :] + float(ls[b])) - 0.5) / 2.0
        assert np.abs(fs[b].cpu().numpy() - exp).max() <= 5e-6
    fr_only = torch.ops.eeg_dcrnn.fft_features(rawb.to(device), 200, 0.0, 1.0, False, None, None)[0]
    assert torch.equal(fr_only, ops.fft_features(rawb.to(device), window=200, mean=0.0, std=1.0)[0])
    for (b, n, w, nwin) in ((1, 1, 4, 1), (2, 32, 252, 2), (3, 19, 200, 60), (70, 19, 8, 5), (1, 1, 200, 1), (5, 3, 200, 7)):   # smallest / widest window, long clip
        rawb = torch.randn(b, n, w * nwin, generator=g) * 10.0
        fr, _ = ops.fft_features(rawb.to(device), window=w, mean=0.0, std=1.0)
        assert fr.shape == (b, nwin, n, w // 2)
        for i in (0, b - 1):
            ref = orc.fft_features(rawb[i].numpy().astype(np.float64), window=w)
            assert np.abs(fr[i].cpu().numpy() - ref).max() <= 5e-6, (b, n, w, nwin)


def check_empty_inputs(device):
    """Empty batches / sequences: the reference raises RuntimeError (model.py:253-255,321-324: its `reshape(..., -1)` of a tensor
    without elements is ambiguous -- probed on the genuine reference: B = 0 and T = 0 alike, both models); so does this package,
    from the module level down to the C ABI ("empty sequence/batch", "empty input"), never a launch with a zero-sized grid."""
    import pytest
    from eeg_gnn_ssl_amd import DCRNNModel_classification, DCRNNModel_nextTimePred, ops
    cfg = orc.DCRNNConfig(filter_type="laplacian", input_dim=8, output_dim=8)
    cls = DCRNNModel_classification(make_args(cfg), 1, device=device).to(device)
    ssl = DCRNNModel_nextTimePred(make_args(cfg), device=device).to(device)
    n = cfg.num_nodes
    for b, t_len in ((0, 4), (2, 0)):
        x = torch.zeros(b, t_len, n, 8, device=device)
        sup = [torch.zeros(b, n, n, device=device)]
        with pytest.raises(RuntimeError):
            cls(x, torch.full((b,), max(t_len, 1), dtype=torch.int64, device=device), sup)
        with pytest.raises(RuntimeError):
            ssl(x, torch.zeros(b, 3, n, 8, device=device), sup)
    with pytest.raises(RuntimeError):                          # no decoder steps
        ssl(torch.zeros(2, 4, n, 8, device=device), torch.zeros(2, 0, n, 8, device=device), [torch.zeros(2, n, n, device=device)])
    # operator level: the C ABI refuses, the message names the cause
    with pytest.raises(RuntimeError, match="empty"):
        ops.fft_features(torch.zeros(0, n, 400, device=device), window=200)
    with pytest.raises(RuntimeError):
        ops.correlation_supports(torch.zeros(0, 4, n, 8, device=device), top_k=3)
    h, m = 64, 3
    wg, bg = torch.zeros((8 + h) * m, 2 * h, device=device), torch.zeros(2 * h, device=device)
    wc, bc = torch.zeros((8 + h) * m, h, device=device), torch.zeros(h, device=device)
    pz = torch.zeros(1, m - 1, n, n, device=device)
    with pytest.raises(RuntimeError, match="empty"):
        torch.ops.eeg_dcrnn.dcgru_layer(torch.zeros(0, 2, n, 8, device=device), 0, None, pz, 0, wg, bg, wc, bc, None, None, n, h, m, 0, False, False, None)
    # every other operator with a zero-sized operand: a RuntimeError from the C ABI's own checks, never a fault and never a launch
    # with an empty grid (the size queries these paths call first used to divide by the batch size: tests/test_abi.py)
    o = torch.ops.eeg_dcrnn
    z = lambda *shape: torch.zeros(*shape, device=device)                      # noqa: E731
    i64 = lambda *shape: torch.zeros(*shape, dtype=torch.int64, device=device)  # noqa: E731
    p4 = pz
    refused = {
        "hop_polys": lambda: o.hop_polys([z(0, n, n)], 2, 0),
        "diffusion_hops": lambda: o.diffusion_hops(z(0, n, 8), p4, 0, 0),
        "dconv": lambda: o.dconv(z(0, n, 8), p4, 0, z(8 * m, 64), z(64)),
        "dconv_bwd": lambda: o.dconv_bwd(z(0, n, 64), z(0, n, 8), p4, 0, z(8 * m, 64), True),
        "gather_last B=0": lambda: o.gather_last(z(4, 0, n * h), i64(0)),
        "gather_last T=0": lambda: o.gather_last(z(0, 2, n * h), i64(2) + 1),
        "cls_head": lambda: o.cls_head(z(0, n, h), z(1, h), z(1), 0.0, None),
        "cls_head_bwd": lambda: o.cls_head_bwd(z(0, n, h), z(1, h), z(0, 1), torch.zeros(0, 1, dtype=torch.int32, device=device), 0.0, None, z(1, h), z(1)),
        "bce_logits": lambda: o.bce_logits(z(0), z(0)),
        "ce_logits": lambda: o.ce_logits(z(0, 4), i64(0)),
        "masked_loss": lambda: o.masked_loss(z(0, 3, n, 8), z(0, 3, n, 8), False, 0.0, 1.0, 0.0, 1),
        "pack_cell": lambda: o.pack_cell(z(h * m, 2 * h), bg, z(h * m, h), bc, 0, h, m),
        "corr_graph T=0": lambda: o.corr_graph(z(2, 0, n, 8), 3),
        "dcgru_layer B=0": lambda: o.dcgru_layer(z(3, 0, n, 8), 0, None, pz, 0, wg, bg, wc, bc, None, None, n, h, m, 0, True, False, None),
        "teacher_flags": lambda: o.teacher_flags_(i64(2), i64(1), 1, 3000.0, 0),
        "clip_adam": lambda: ops.clip_adam_step_dev(z(0), z(0), z(0), z(0), torch.zeros(1, dtype=torch.int32, device=device), z(1),
                                                    (0.9, 0.999), 1e-8, 0.0, 5.0, 1.0, z(64), z(1)),
    }
    for name, call in refused.items():
        with pytest.raises(RuntimeError):
            call()
            pytest.fail(f"{name}: accepted an empty operand")
    assert o.dropout_mask(i64(2), 0, 0.5).numel() == 0                          # (an empty mask is a valid answer)


def check_malformed_inputs(device):
    """Operands whose extents do not fit each other are refused by the host layer BEFORE a kernel indexes them (the kernels take
    extents from the dims struct, not from the tensors): lengths of another batch size, supports / hop polynomials of another node
    count or batch, a decoder state of another layer count.  Found by probing at the end of round 5: `seq_lengths` with one
    entry for two clips and supports of 18 nodes for a 19-node model were read out of bounds.  Lenient where the reference is
    (integer / float lengths of any dtype, (B,1) lengths, float64 supports, flattened node x feature inputs, shared (N,N) supports)."""
    import pytest
    from eeg_gnn_ssl_amd import DCRNNModel_classification, DCRNNModel_nextTimePred, ops
    cfg = orc.DCRNNConfig(filter_type="dual_random_walk", input_dim=8, output_dim=8)
    torch.manual_seed(0)
    cls = DCRNNModel_classification(make_args(cfg), 1, device=device).to(device)
    ssl = DCRNNModel_nextTimePred(make_args(cfg), device=device).to(device)
    b, t_len, n = 2, 3, cfg.num_nodes
    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, t_len, n, 8, generator=g).to(device)
    lens = torch.full((b,), t_len, dtype=torch.int64, device=device)
    sup = [torch.rand(b, n, n, generator=g).to(device) for _ in range(2)]
    good = cls(x, lens, sup).detach()
    for what, call in {
        "lengths as int32": lambda: cls(x, lens.int(), sup),
        "lengths as float": lambda: cls(x, lens.float(), sup),
        "lengths (B,1)": lambda: cls(x, lens.view(b, 1), sup),
        "float64 supports": lambda: cls(x, lens, [s.double() for s in sup]),
        "flattened node x feature inputs": lambda: cls(x.reshape(b, t_len, n * 8), lens, sup),
        "non-contiguous inputs": lambda: cls(torch.stack([x, x], dim=-1)[..., 0], lens, sup),
    }.items():
        assert torch.equal(call().detach(), good), what
    shared = cls(x, lens, [sup[0][0], sup[1][0]]).detach()                     # (N,N) supports = the same graph for every clip
    assert torch.allclose(shared[0], good[0], atol=1e-6)
    o = torch.ops.eeg_dcrnn
    z = lambda *shape: torch.zeros(*shape, device=device)                       # noqa: E731
    h, m = cfg.rnn_units, 5
    p_ok, _ = ops.hop_polys(sup, 2, b)
    wg, bg, wc, bc = z((8 + h) * m, 2 * h), z(2 * h), z((8 + h) * m, h), z(h)
    layer = lambda xx, pp, ll=None, h0=None: o.dcgru_layer(xx, 0, h0, pp, 1, wg, bg, wc, bc, ll, None, n, h, m, 0, False, True, None)   # noqa: E731
    layer(x.transpose(0, 1).contiguous(), p_ok)                                 # (the well-formed call goes through)
    refused = {
        "lengths of another batch size": lambda: cls(x, lens[:1], sup),
        "supports of another batch size": lambda: cls(x, lens, [torch.rand(3, n, n).to(device)] * 2),
        "supports with a batch of 1": lambda: cls(x, lens, [s[:1] for s in sup]),
        "one batched and one (1,N,N) support": lambda: cls(x, lens, [sup[0], sup[1][:1]]),
        "supports of 18 nodes": lambda: cls(x, lens, [torch.rand(b, 18, 18).to(device)] * 2),
        "too few supports": lambda: cls(x, lens, sup[:1]),
        "too many supports": lambda: cls(x, lens, sup + sup[:1]),
        "inputs of 18 nodes": lambda: cls(torch.randn(b, t_len, 18, 8).to(device), lens, sup),
        "inputs of another width": lambda: cls(torch.randn(b, t_len, n, 12).to(device), lens, sup),
        "float64 inputs": lambda: cls(x.double(), lens, sup),
        "targets of another batch": lambda: ssl(x, torch.randn(3, 2, n, 8).to(device), sup),
        "targets of 18 nodes": lambda: ssl(x, torch.randn(b, 2, 18, 8).to(device), sup),
        "layer: P of another node count": lambda: layer(x.transpose(0, 1).contiguous(), z(b, m - 1, 18, 18)),
        "layer: P of another batch": lambda: layer(x.transpose(0, 1).contiguous(), z(3, m - 1, n, n)),
        "layer: P of another hop count": lambda: layer(x.transpose(0, 1).contiguous(), z(b, 2, n, n)),
        "layer: inputs of 18 nodes": lambda: layer(z(t_len, b, 18, 8), p_ok),
        "layer: short lengths": lambda: layer(x.transpose(0, 1).contiguous(), p_ok, lens[:1]),
        "layer: h0 of another batch": lambda: layer(x.transpose(0, 1).contiguous(), p_ok, None, z(3, n * h)),
        "gather_last: short lengths": lambda: o.gather_last(z(t_len, b, n * h), lens[:1]),
        "dconv: P of another node count": lambda: o.dconv(z(b, n, 8), z(b, m - 1, 18, 18), 1, z(8 * m, 64), z(64)),
        "dconv: weight rows": lambda: o.dconv(z(b, n, 8), p_ok, 1, z(8 * 3, 64), z(64)),
        "diffusion_hops: P of another batch": lambda: o.diffusion_hops(z(b, n, 8), z(3, m - 1, n, n), 1, b),
    }
    for what, call in refused.items():
        with pytest.raises(RuntimeError):
            call()
            pytest.fail(f"{what}: accepted")
    # index operands with values outside their range never address other memory: a class label outside 0..C-1 gives a NaN loss
    # (torch: a device-side assert that ends the process) with the softmax as that clip's gradient; a reflection "permutation"
    # entry outside 0..N-1 falls back to the node itself
    lg = torch.randn(3, 4, generator=g).to(device)
    loss, dl = o.ce_logits(lg, torch.tensor([1, 7, -3], device=device))
    assert torch.isnan(loss) and torch.isfinite(dl).all()
    assert torch.allclose(dl[1:], torch.softmax(lg[1:], dim=1) / 3, atol=1e-6)
    raw = torch.randn(2, n, 400, generator=g).to(device)
    ident = torch.arange(n, dtype=torch.int32).repeat(2, 1).to(device)
    bad = ident.clone()
    bad[0, 3], bad[1, 7] = 99, -5
    f_id = ops.fft_features(raw, window=200, mean=0.0, std=1.0, perm=ident)
    f_bad = ops.fft_features(raw, window=200, mean=0.0, std=1.0, perm=bad)
    assert torch.equal(f_id[0], f_bad[0]) and torch.equal(f_id[1], f_bad[1])
    for what, call in {
        "bce: fewer targets than logits": lambda: o.bce_logits(z(4), z(3)),
        "ce: fewer targets than rows": lambda: o.ce_logits(z(4, 4), torch.zeros(3, dtype=torch.int64, device=device)),
        "ce: 1-D logits": lambda: o.ce_logits(z(4), torch.zeros(4, dtype=torch.int64, device=device)),
        "fft_features: perm of another shape": lambda: ops.fft_features(raw, window=200, perm=ident[:1]),
        "fft_features: log_scale of another batch": lambda: ops.fft_features(raw, window=200, mean=0.0, std=1.0, log_scale=z(3)),
        "cls_head: fc.weight of another width": lambda: o.cls_head(z(b, n, h), z(1, 32), z(1), 0.0, None),
        "cls_head_bwd: arg of another shape": lambda: o.cls_head_bwd(z(b, n, h), z(1, h), z(b, 1), torch.zeros(b, 2, dtype=torch.int32, device=device),
                                                                     0.0, None, z(1, h), z(1)),
    }.items():
        with pytest.raises(RuntimeError):
            call()
            pytest.fail(f"{what}: accepted")
    # a decoder state of another layer count / batch
    dec = ssl.decoder
    with pytest.raises(RuntimeError):
        dec(z(2, b, n, 8), z(3, b, n * h), sup)
    with pytest.raises(RuntimeError):
        dec(z(2, b, n, 8), z(2, 3, n * h), sup)


def check_raw_input_chain(device, b=4, t_len=3):
    """Raw signals in: TrainStep(raw_window=200) runs the reference's DataLoader-side chain on the device in front of the model
    -- log|FFT| per 1-s step (data_utils.py:13-35), z-score (utils.py:393-428), per-clip correlation graph from the
    UN-standardised features (dataloader_detection.py:258-307,346-354) -- and the detection step.  Against the oracle chain
    (numpy FFT -> host graph builders -> oracle model) on the same signals: loss and every parameter gradient."""
    from eeg_gnn_ssl_amd import DCRNNModel_classification, utils
    from eeg_gnn_ssl_amd.train_step import TrainStep
    g = torch.Generator().manual_seed(31)
    raw = 20.0 * torch.randn(b, 19, t_len * 200, generator=g)
    y = (torch.rand(b, generator=g) > 0.5).float()
    lengths = torch.full((b,), t_len, dtype=torch.int64)
    mean, std = 5.53, 0.65
    cfg = orc.DCRNNConfig(filter_type="dual_random_walk", num_classes=1)
    params = orc.init_params(cfg, "classification", seed=2)
    model = DCRNNModel_classification(make_args(cfg), 1, device=device)
    load(model, params, device)
    model.train()
    st = TrainStep(model, task="detection", raw_window=200, raw_mean=mean, raw_std=std)
    loss = st.forward_backward(raw.to(device), y.to(device), lengths.to(device), None)
    feats = np.stack([orc.fft_features(raw[i].numpy().astype(np.float64), window=200) for i in range(b)])      # (B, T, N, 100)
    x = torch.from_numpy(((feats - mean) / std).astype(np.float32))
    s1, s2 = [], []
    for i in range(b):
        sp = utils.compute_supports(utils.correlation_graph(feats[i], top_k=3), "dual_random_walk")
        s1.append(sp[0]); s2.append(sp[1])
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    lo = orc.bce_with_logits(orc.classification_forward(po, cfg, x, lengths, [torch.stack(s1), torch.stack(s2)]), y)
    lo.backward()
    assert abs(float(loss.item()) - float(lo.item())) < 2e-5, (float(loss.item()), float(lo.item()))
    for k, q in model.named_parameters():
        assert_close_scaled(q.grad.cpu().numpy(), po[k].grad.numpy(), f"raw chain/d_{k}", tol=1e-4)


def check_cls_head_loss(device, shapes=((1, 19, 64, 1), (5, 19, 64, 4), (256, 19, 64, 1), (37, 21, 32, 3))):
    """`ops.cls_head_loss` (head + criterion + the head's backward, two launches) against the launch-by-launch public path --
    `ops.cls_head` -> `bce_logits` / `ce_logits` -> autograd through the head: logits, dlogits and dz are the SAME arithmetic
    (bit-equal), loss / dW / dbias are sums over the batch in another fixed order (rounding-level agreement), with and without
    dropout (same generator pair), for batches that do not fill the last workgroup; gradients land in the GradSink buffers or
    accumulate into .grad; twice the same call is bit-identical."""
    from eeg_gnn_ssl_amd import ops
    g = torch.Generator().manual_seed(77)
    for (b, n, h, c) in shapes:
        for p_drop in (0.0, 0.5):
            z = torch.randn(b, n, h, generator=g).to(device)
            w = (0.3 * torch.randn(c, h, generator=g)).to(device).requires_grad_(True)
            bias = (0.1 * torch.randn(c, generator=g)).to(device).requires_grad_(True)
            task = "detection" if c == 1 else "classification"
            y = ((torch.rand(b, generator=g) > 0.5).float() if c == 1 else torch.randint(0, c, (b,), generator=g)).to(device)
            st1 = torch.tensor([4242, 11], dtype=torch.int64, device=device)
            st2 = st1.clone()
            # launch by launch
            zr = z.clone().requires_grad_(True)
            logits, used = ops.cls_head(zr, w, bias, p_drop, st1, return_rng_used=True)
            loss_ref, seed = (torch.ops.eeg_dcrnn.bce_logits(logits.detach().view(-1), y) if c == 1
                              else torch.ops.eeg_dcrnn.ce_logits(logits.detach(), y))
            logits.backward(seed.view_as(logits))
            # fused
            w2, b2 = w.detach().clone().requires_grad_(True), bias.detach().clone().requires_grad_(True)
            loss, lg, dz = ops.cls_head_loss(z, w2, b2, y, task, p_drop, st2)
            assert st1.tolist() == st2.tolist()
            assert torch.equal(lg, logits.detach()), (b, c, p_drop)
            assert torch.equal(dz, zr.grad), (b, c, p_drop, float((dz - zr.grad).abs().max()))
            assert abs(float(loss) - float(loss_ref)) <= 2e-6 * max(1.0, abs(float(loss_ref)))
            for got, want, nm in ((w2.grad, w.grad, "dW"), (b2.grad, bias.grad, "db")):
                assert float((got - want).abs().max()) <= 2e-6 * max(1e-3, float(want.abs().max())), (nm, b, c, p_drop)
            # again into .grad: accumulates; bit-identical contribution
            st3 = torch.tensor([4242, 11], dtype=torch.int64, device=device)
            first = w2.grad.clone()
            loss_b, _, dz_b = ops.cls_head_loss(z, w2, b2, y, task, p_drop, st3)
            assert torch.equal(dz_b, dz) and float(loss_b) == float(loss) and torch.equal(w2.grad, first + first)
    # refusals
    z = torch.randn(2, 19, 64).to(device)
    w = torch.randn(4, 64).to(device)
    for bad, msg in ((lambda: ops.cls_head_loss(z, w, torch.zeros(4).to(device), torch.zeros(3, dtype=torch.int64).to(device), "classification"), "targets"),
                     (lambda: ops.cls_head_loss(z, w, torch.zeros(4).to(device), torch.zeros(2).to(device), "detection"), "one logit"),
                     (lambda: ops.cls_head_loss(z, w, torch.zeros(4).to(device), torch.zeros(2, dtype=torch.int64).to(device), "classification", 0.5), "generator")):
        try:
            bad()
        except RuntimeError as e:
            assert msg in str(e), (msg, str(e))
        else:
            raise AssertionError(f"cls_head_loss accepted operands that do not fit ({msg})")
    lo, _, _ = ops.cls_head_loss(z, w, torch.zeros(4).to(device), torch.tensor([0, 7]).to(device), "classification")
    assert bool(torch.isnan(lo))                               # a label outside 0..C-1: NaN loss, no out-of-bounds read


def check_fused_head_step_equals_public_path(device, adj3d, task="detection"):
    """TrainStep with the fused head (default) against the same step through model.forward -> loss kernel -> autograd
    (fused_head = False): loss and the whole flat gradient agree at rounding level; with dropout the same masks are drawn."""
    from eeg_gnn_ssl_amd import DCRNNModel_classification
    from eeg_gnn_ssl_amd.train_step import TrainStep
    nc = 1 if task == "detection" else 4
    cfg = orc.DCRNNConfig(filter_type="laplacian", num_classes=nc, input_dim=8, rnn_units=32)
    g = torch.Generator().manual_seed(5)
    b, t_len = 7, 3
    x = torch.randn(b, t_len, 19, 8, generator=g)
    y = (torch.rand(b, generator=g) > 0.5).float() if nc == 1 else torch.randint(0, nc, (b,), generator=g)
    lengths = torch.randint(1, t_len + 1, (b,), generator=g)
    sup = [t.to(device) for t in cases.supports_for("laplacian", adj3d, b)]
    grads, losses = [], []
    for fused in (True, False):
        params = orc.init_params(cfg, "classification", seed=9)
        args = make_args(cfg)
        args.dropout = 0.4
        model = DCRNNModel_classification(args, nc, device=device)
        load(model, params, device)
        model.train()
        model.set_dropout_seed(99, 0)
        st = TrainStep(model, task=task)
        st.fused_head = fused
        losses.append(float(st.forward_backward(x.to(device), y.to(device), lengths.to(device), sup)))
        grads.append(st.fp.flat_grad.clone())
        assert model.dropout_rng_state() == (99, b * 19 * 32 // 4)
    assert abs(losses[0] - losses[1]) <= 2e-6 * max(1.0, abs(losses[1])), losses
    assert float((grads[0] - grads[1]).abs().max()) <= 2e-6 * float(grads[1].abs().max())


def expected_augmentation(seed, offset, batch, swap_perm):
    """the documented function of the generator pair (include/eeg_dcrnn.h eeg_dcrnn_augment_draw): clip b takes Philox counter
    offset + b; reflection coin = top bit of word 0, scale = 0.8 + 0.4 * word 1 / 2^32"""
    words = philox4x32_10_numpy(np.uint64(offset) + np.arange(batch, dtype=np.uint64), int(seed))
    flags = (words[:, 0] >> np.uint32(31)).astype(np.int32)
    scale = 0.8 + 0.4 * (words[:, 1].astype(np.float64) / 4294967296.0)
    ident = np.arange(len(swap_perm), dtype=np.int32)
    perm = np.where(flags[:, None] == 1, np.asarray(swap_perm, dtype=np.int32)[None, :], ident[None, :])
    return flags, perm, np.log(scale).astype(np.float32)


def check_augmentation_draws(device, adj3d):
    """known-answer test of the device-side augmentation draws (dataloader_detection.py:233-256): flags / perm / log_scale are the
    documented function of the generator pair, the generator advances by one counter per clip on the stream, the per-clip supports
    are the plain or the reflected set, the coin is fair and the scale uniform in [0.8, 1.2)."""
    from eeg_gnn_ssl_amd import ops, utils
    sp = utils.swap_permutation(19)
    assert sorted(sp.tolist()) == list(range(19)) and sp[sp.long()].tolist() == list(range(19))      # an involution
    st = torch.tensor([123456789123, 7], dtype=torch.int64, device=device)
    plain = utils.compute_supports(adj3d, "dual_random_walk")
    refl = utils.reflected_supports(adj3d, "dual_random_walk")
    flags, perm, ls, sel = ops.draw_augmentation(st, 10, sp.to(device), [t.to(device) for t in plain], [t.to(device) for t in refl])
    ef, ep, el = expected_augmentation(123456789123, 7, 10, sp.numpy())
    assert flags.dtype == torch.int32 and flags.tolist() == ef.tolist() and perm.cpu().numpy().tolist() == ep.tolist()
    np.testing.assert_allclose(ls.cpu().numpy(), el, rtol=0, atol=2e-7)
    assert st.tolist() == [123456789123, 17]
    assert len(sel) == 2 and tuple(sel[0].shape) == (10, 19, 19)
    for i in range(2):
        for b in range(10):
            want = refl[i] if ef[b] else plain[i]
            assert torch.equal(sel[i][b].cpu(), want), (i, b)
    f2, p2, l2, none = ops.draw_augmentation(st, 4096, sp.to(device))
    assert none is None and st.tolist() == [123456789123, 17 + 4096]
    ef2, _, el2 = expected_augmentation(123456789123, 17, 4096, sp.numpy())
    assert f2.tolist() == ef2.tolist()
    rate = float(f2.float().mean().item())
    assert abs(rate - 0.5) < 3 * 0.5 / np.sqrt(4096.0), rate
    sc = torch.exp(l2.double()).cpu().numpy()
    assert sc.min() >= 0.8 - 1e-6 and sc.max() < 1.2 + 1e-6 and abs(sc.mean() - 1.0) < 0.01, (sc.min(), sc.max(), sc.mean())
    try:
        ops.draw_augmentation(st, 4, sp.to(device), [plain[0].to(device)], None)
    except RuntimeError as e:
        assert "BOTH" in str(e)
    else:
        raise AssertionError("a plain set without its reflected partner must be refused")


def check_augmented_step(device, adj3d, graph="distance", raw=True, b=6, t_len=2):
    """TrainStep(data_augment=True): the reference's per-sample augmentation (dataloader_detection.py:384-393: `_random_reflect`,
    `_random_scale`, then the scaler) with the draws made on the device, and its graph side (:402-409): the distance graph of a
    reflected clip is `_get_combined_graph(swap_nodes)` (pinned by golden_reflect_v1.npz), the correlation graph is built from the
    UN-reflected, un-scaled clip (Q10).  The draws are read back and handed to the oracle chain (numpy FFT -> reflect -> + log
    scale -> z-score -> host graph builders -> oracle model): loss and every parameter gradient agree.  raw=False: the same on
    already standardised features (x[b, :, perm[b]] + log(scale) / std)."""
    from eeg_gnn_ssl_amd import DCRNNModel_classification, utils
    from eeg_gnn_ssl_amd.train_step import TrainStep
    g = torch.Generator().manual_seed(57)
    raw_sig = 20.0 * torch.randn(b, 19, t_len * 200, generator=g)
    y = (torch.rand(b, generator=g) > 0.5).float()
    lengths = torch.full((b,), t_len, dtype=torch.int64)
    mean, std = 5.53, 0.65
    filt = "laplacian" if graph == "distance" else "dual_random_walk"
    cfg = orc.DCRNNConfig(filter_type=filt, num_classes=1)
    params = orc.init_params(cfg, "classification", seed=3)
    model = DCRNNModel_classification(make_args(cfg), 1, device=device)
    load(model, params, device)
    model.train()
    feats = np.stack([orc.fft_features(raw_sig[i].numpy().astype(np.float64), window=200) for i in range(b)])      # (B, T, N, 100)
    plain = utils.compute_supports(adj3d, filt)
    refl = utils.reflected_supports(adj3d, filt)
    kw = dict(raw_window=200, raw_mean=mean, raw_std=std) if raw else dict(feature_std=std)
    st = TrainStep(model, task="detection", data_augment=True, reflected_supports=refl if graph == "distance" else None, **kw)
    sup_in = [p_.unsqueeze(0).repeat(b, 1, 1).to(device) for p_ in plain] if graph == "distance" else None      # batched copies, as the trainers pass them
    x_in = raw_sig if raw else torch.from_numpy(((feats - mean) / std).astype(np.float32))
    draws = []
    for _ in range(2):                                           # two steps: the generator advanced, the draws differ
        loss = st.forward_backward(x_in.to(device), y.to(device), lengths.to(device), sup_in)
        flags, perm, ls = (t.cpu() for t in st.last_augmentation)
        draws.append(flags.tolist() + ls.tolist())
        fa = np.stack([feats[i][:, perm[i].numpy(), :] + float(ls[i]) for i in range(b)])
        x = torch.from_numpy(((fa - mean) / std).astype(np.float32))
        if graph == "distance":
            sups = [torch.stack([(refl[k] if flags[i] else plain[k]) for i in range(b)]) for k in range(len(plain))]
        else:
            src = feats if raw else (feats - mean) / std            # (feature inputs: the graph of the un-augmented INPUT)
            per = [utils.compute_supports(utils.correlation_graph(src[i], top_k=3), filt) for i in range(b)]
            sups = [torch.stack([per[i][k] for i in range(b)]) for k in range(2)]
        po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        lo = orc.bce_with_logits(orc.classification_forward(po, cfg, x, lengths, sups), y)
        lo.backward()
        assert abs(float(loss.item()) - float(lo.item())) < 2e-5, (float(loss.item()), float(lo.item()))
        for k, q in model.named_parameters():
            assert_close_scaled(q.grad.cpu().numpy(), po[k].grad.numpy(), f"augmented step {graph}/d_{k}", tol=1e-4)
    assert draws[0] != draws[1]
    assert 0 < sum(draws[0][:b]) + sum(draws[1][:b]) < 2 * b       # both outcomes of the coin were exercised
    model.eval()                                                 # no augmentation outside training
    st.last_augmentation = None
    st.forward_backward(x_in.to(device), y.to(device), lengths.to(device), sup_in)
    assert st.last_augmentation is None


def check_split_bf16(device, adj3d, filt="laplacian", din=100, layers=2, t_len=3, b=3, seed=4):
    """The OPT-IN three-term bf16 split of the hoisted NN GEMMs (ops.set_gemm_mode(1); include/eeg_dcrnn.h eeg_layer_dims.pack3):
    logits and every parameter gradient of the classification model against the oracle at the suite's tolerance -- the split keeps
    fp32-level accuracy -- and a result that differs from the fp32-MFMA path in the last bits (proof that the other kernels ran),
    bit-identical between two runs; rnn_units != 64 keeps the fp32 kernels."""
    from eeg_gnn_ssl_amd import DCRNNModel_classification, ops
    if ops.GEMM_MODE != 0:
        import pytest
        pytest.skip("the suite itself runs with EEG_DCRNN_SPLIT_BF16=1: this test compares the two modes")
    cfg = orc.DCRNNConfig(filter_type=filt, input_dim=din, rnn_units=64, num_rnn_layers=layers, num_classes=4)
    sup = cases.supports_for(filt, adj3d, b)
    while True:
        g = torch.Generator().manual_seed(seed)
        params = orc.init_params(cfg, "classification", seed=seed)
        x = torch.randn(b, t_len, 19, din, generator=g)
        seq = torch.randint(max(1, t_len // 2), t_len + 1, (b,), generator=g)
        y = torch.randint(0, 4, (b,), generator=g)
        po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        # the head is fc(relu(h)) followed by a max over nodes: an input of the ReLU within rounding distance of 0 at a maximising node
        # (or two nodes tied for the maximum) has two valid sub-gradients, and which one an implementation takes is decided by its
        # summation order ("ReLU kink").  Such a draw says nothing about the GEMMs: take the next seed.
        with torch.no_grad():
            h0 = torch.zeros(layers, b, 19 * 64)
            _, top = orc.encoder_forward(params, cfg, x.transpose(0, 1), h0, sup)
            last = orc.last_relevant(top.transpose(0, 1), seq).view(b, 19, 64)
            nl = torch.relu(last) @ params["fc.weight"].t() + params["fc.bias"]
            srt = nl.sort(dim=1, descending=True).values
            arg = nl.argmax(dim=1)                                              # (B, C) maximising nodes
            zmin = last.abs().gather(1, arg.unsqueeze(-1).expand(-1, -1, 64)).min().item()
            tie = (srt[:, 0] - srt[:, 1]).min().item() if srt.shape[1] > 1 else 1.0
        if zmin > 1e-5 and tie > 1e-4:
            break
        seed += 1
    lo = orc.classification_forward(po, cfg, x, seq, sup)
    orc.cross_entropy(lo, y).backward()

    def run():
        model = DCRNNModel_classification(make_args(cfg), 4, device=device)
        load(model, params, device)
        model.train()
        lg = model(x.to(device), seq.to(device), [s.to(device) for s in sup])
        torch.nn.functional.cross_entropy(lg, y.to(device)).backward()
        return lg.detach().cpu(), {k: q.grad.detach().cpu().clone() for k, q in model.named_parameters()}

    assert ops.GEMM_MODE == 0
    lg32, gr32 = run()
    prev = ops.set_gemm_mode(1)
    try:
        assert ops._pack3_halves(din, 64, cfg.num_matrices) > 0 and ops._pack3_halves(din, 32, cfg.num_matrices) == 0
        lg3, gr3 = run()
        lg3b, gr3b = run()
    finally:
        ops.set_gemm_mode(prev)
    assert_close(lg3.numpy(), lo.detach().numpy(), "split-bf16 logits vs oracle")
    for k in gr3:
        assert_close_scaled(gr3[k].numpy(), po[k].grad.numpy(), f"split-bf16 d_{k} vs oracle")
        assert torch.equal(gr3[k], gr3b[k]), k
    assert torch.equal(lg3, lg3b)
    differs = any(not torch.equal(gr3[k], gr32[k]) for k in gr3)
    assert differs, "the bf16 split produced bit-identical gradients to the fp32 path: it did not run"
    worst = max(float((gr3[k] - gr32[k]).abs().max() / gr32[k].abs().max().clamp_min(1e-12)) for k in gr3)
    assert worst < 2e-5, worst           # fp32-level agreement between the two arithmetic modes


def check_torch_ops(device, adj3d, opcheck_utils=("test_schema", "test_autograd_registration", "test_faketensor")):
    """The operators are registered with the PyTorch dispatcher (north_star: "exposed as a torch.ops extension"):
    `torch.ops.eeg_dcrnn.*` called DIRECTLY (no module, no Python wrapper) against the oracle, and run through
    `torch.library.opcheck` (schema / aliasing, autograd registration, fake-tensor implementation)."""
    import torch.library
    from eeg_gnn_ssl_amd import ops  # noqa: F401  (registers the library)
    E = torch.ops.eeg_dcrnn
    g = torch.Generator().manual_seed(17)
    n, h, din, t_len, b, k = 19, 16, 8, 4, 3, 2
    cfg = orc.DCRNNConfig(filter_type="dual_random_walk", input_dim=din, rnn_units=h, num_rnn_layers=1, num_classes=1,
                          max_diffusion_step=k)
    params = orc.init_params(cfg, "classification", seed=3)
    pre = "encoder.encoding_cells.0."
    wg, bg, wc, bc = (params[pre + s].clone() for s in ("dconv_gate.weight", "dconv_gate.biases", "dconv_candidate.weight",
                                                         "dconv_candidate.biases"))
    bg, bc = bg + 0.1, bc - 0.05
    sup = cases.supports_for("dual_random_walk", adj3d, b)
    x = torch.randn(t_len, b, n, din, generator=g)
    h0 = 0.3 * torch.randn(b, n * h, generator=g)
    up = torch.randn(t_len, b, n * h, generator=g)
    # oracle: one encoder layer with an initial state, gradients of sum(hseq * up)
    po = {pre + "dconv_gate.weight": wg, pre + "dconv_gate.biases": bg, pre + "dconv_candidate.weight": wc,
          pre + "dconv_candidate.biases": bc}
    po = {kk: v.clone().requires_grad_(True) for kk, v in po.items()}
    xo, h0o = x.clone().requires_grad_(True), h0.clone().requires_grad_(True)
    _, top = orc.encoder_forward(po, cfg, xo, h0o.unsqueeze(0), sup)
    (top * up).sum().backward()
    # the operators, directly
    d = lambda t: t.to(device)   # noqa: E731
    P = E.hop_polys([d(s) for s in sup], k, b)
    assert tuple(P.shape) == (b, 2 * k, n, n)
    leaf = [d(t).clone().requires_grad_(True) for t in (x, h0, wg, bg, wc, bc)]
    xd, h0d, wgd, bgd, wcd, bcd = leaf
    hext, hsel, saved = E.dcgru_layer(xd, 0, h0d, P, 1, wgd, bgd, wcd, bcd, None, None, n, h, 2 * k + 1, 0, True, True, None)
    assert tuple(hext.shape) == (t_len + 1, b, n * h) and len(saved) == 10
    assert_close(hext[1:].detach().cpu().numpy(), top.detach().numpy(), "torch.ops dcgru_layer hseq")
    assert_close(hsel.detach().cpu().numpy(), top[-1].detach().numpy(), "torch.ops dcgru_layer hsel")
    assert_close(hext[0].detach().cpu().numpy(), h0.numpy(), "hext slot 0 = initial state", tol=1e-7)
    (hext[1:] * d(up)).sum().backward()
    for got, ref, nm in ((xd.grad, xo.grad, "dx"), (h0d.grad, h0o.grad, "dh0"), (wgd.grad, po[pre + "dconv_gate.weight"].grad, "dWg"),
                         (bgd.grad, po[pre + "dconv_gate.biases"].grad, "dbg"), (wcd.grad, po[pre + "dconv_candidate.weight"].grad, "dWc"),
                         (bcd.grad, po[pre + "dconv_candidate.biases"].grad, "dbc")):
        assert_close_scaled(got.cpu().numpy(), ref.numpy(), f"torch.ops dcgru_layer {nm}")
    # hop planes handed to a second layer == that layer diffusing its input itself
    w2 = [d(t) for t in (0.1 * torch.randn((h + h) * 5, 2 * h, generator=g), torch.zeros(2 * h),
                         0.1 * torch.randn((h + h) * 5, h, generator=g), torch.zeros(h))]
    xin = hext.detach().view(t_len + 1, b, n, h)
    a1 = E.dcgru_layer(xin, 1, None, P, 1, *w2, None, saved[7].detach(), n, h, 5, 0, False, True, None)[0]
    a2 = E.dcgru_layer(xin, 1, None, P, 1, *w2, None, None, n, h, 5, 0, False, True, None)[0]
    assert_close(a1.cpu().numpy(), a2.cpu().numpy(), "x_planes hand-over", tol=1e-6)
    # opcheck: schema + aliasing annotations, autograd registration, fake (meta) implementations
    nog = [t.detach() for t in leaf]
    samples = [
        (E.hop_polys.default, ([d(s) for s in sup], k, b)),
        (E.pack_cell.default, (nog[2], nog[3], nog[4], nog[5], din, h, 5)),
        (E.diffusion_hops.default, (nog[0].reshape(t_len * b, n, din), P, 1, b)),
        (E.dcgru_layer.default, (xd.detach().requires_grad_(True), 0, h0d.detach().requires_grad_(True), P, 1,
                                 *[t.detach().requires_grad_(True) for t in (wgd, bgd, wcd, bcd)], None, None, n, h, 5, 0, True, True, None)),
        (E.dconv.default, (torch.randn(b, n, din + h, generator=g).to(device).requires_grad_(True), P, 1,
                           nog[2].clone().requires_grad_(True), nog[3].clone().requires_grad_(True))),
        (E.cls_head.default, (torch.randn(b, n, h, generator=g).to(device).requires_grad_(True),
                              torch.randn(4, h, generator=g).to(device).requires_grad_(True),
                              torch.randn(4, generator=g).to(device).requires_grad_(True), 0.0, None)),
        # training-mode dropout: the functional head takes the {seed, offset} pair that the (mutating) rng_take_ handed out
        (E.cls_head.default, (torch.randn(b, n, h, generator=g).to(device).requires_grad_(True),
                              torch.randn(4, h, generator=g).to(device).requires_grad_(True),
                              torch.randn(4, generator=g).to(device).requires_grad_(True), 0.5,
                              d(torch.tensor([1234567, 40], dtype=torch.int64)))),
        (E.rng_take_.default, (d(torch.tensor([99, 0], dtype=torch.int64)), 12)),
        (E.dropout_mask.default, (d(torch.tensor([99, 8], dtype=torch.int64)), 50, 0.25)),
        (E.bce_logits.default, (torch.randn(b, generator=g).to(device).requires_grad_(True), d(torch.tensor([1.0, 0.0, 1.0])))),
        (E.ce_logits.default, (torch.randn(b, 4, generator=g).to(device).requires_grad_(True), d(torch.tensor([1, 0, 3])))),
        (E.masked_loss.default, (torch.randn(b, 7, generator=g).to(device).requires_grad_(True), d(torch.randn(b, 7, generator=g)),
                                 True, 0.5, 2.0, 0.0, 1)),
        (E.corr_graph.default, (d(torch.randn(b, t_len, n, 8, generator=g)), 3)),
        (E.gather_last.default, (d(torch.randn(t_len, b, 5, generator=g)), d(torch.tensor([4, 1, 2])))),
        (E.clip_adam_.default, (d(torch.randn(64, generator=g)), d(torch.randn(64, generator=g)), d(torch.zeros(64)), d(torch.zeros(64)),
                                1, 1e-3, 0.9, 0.999, 1e-8, 5e-4, 5.0, 1.0, d(torch.zeros(64)), d(torch.zeros(1)))),
        (E.clip_adam_dev_.default, (d(torch.randn(64, generator=g)), d(torch.randn(64, generator=g)), d(torch.zeros(64)), d(torch.zeros(64)),
                                    d(torch.zeros(1, dtype=torch.int32)), d(torch.full((1,), 1e-3)), 0.9, 0.999, 1e-8, 5e-4, 5.0, 1.0,
                                    d(torch.zeros(64)), d(torch.zeros(1)))),
        (E.teacher_flags_.default, (d(torch.tensor([99, 0], dtype=torch.int64)), d(torch.tensor([120], dtype=torch.int64)), 8, 50.0, 12)),
    ]
    # shapes where the fake implementations used to drift from the library: a 64-unit cell (quad packs bxq / bxtq exist from
    # 64 units on) and a transposed batch-major input at a width the zero-copy row map does not cover (Fin = 8: the library
    # keeps a time-major copy, eeg_dcrnn_batch_major_ok = 1)
    rows64 = (100 + 64) * 3
    samples.append((E.pack_cell.default, (d(0.1 * torch.randn(rows64, 128, generator=g)), d(torch.zeros(128)),
                                          d(0.1 * torch.randn(rows64, 64, generator=g)), d(torch.zeros(64)), 100, 64, 3)))
    x_bm = d(torch.randn(b, t_len, n, din, generator=g))
    samples.append((E.dcgru_layer.default, (x_bm.transpose(0, 1).requires_grad_(True), 0, None, P, 1,
                                            *[t.detach().requires_grad_(True) for t in (wgd, bgd, wcd, bcd)], None, None, n, h, 5, 0, True, False, None)))
    for op, args in samples:
        res = torch.library.opcheck(op, args, test_utils=list(opcheck_utils), raise_exception=True)
        assert all(v == "SUCCESS" for v in res.values()), (str(op), res)


def check_batch_major_input(device, adj3d):
    """The batch-major (B,T,N,Din) model input is consumed WITHOUT a time-major copy where the library allows it
    (eeg_dcrnn_batch_major_ok == 2: diffusion kernel and hoisted GEMMs address it through a (b,t) row map):
    same logits and gradients, bit for bit, as when the encoder is handed a contiguous time-major tensor."""
    from eeg_gnn_ssl_amd import DCRNNModel_classification, _lib
    from eeg_gnn_ssl_amd._lib import LayerDims
    import ctypes
    g = torch.Generator().manual_seed(8)
    shapes = (("laplacian", 3, 5, 3), ("dual_random_walk", 2, 9, 5)) if device != "cpu" else \
        (("laplacian", 2, 4, 3), ("dual_random_walk", 2, 4, 5))          # the emulator runs one lane at a time
    for filt, b, t_len, m in shapes:
        dims = LayerDims(t_len, b, 19, 64, 100, m, 0, 1)
        assert _lib.get_lib().query("eeg_dcrnn_batch_major_ok", ctypes.byref(dims)) == 2
        cfg = orc.DCRNNConfig(filter_type=filt, num_classes=4)
        torch.manual_seed(4)
        model = DCRNNModel_classification(make_args(cfg), 4, device=device).to(device).train()
        x = torch.randn(b, t_len, 19, 100, generator=g).to(device)
        lengths = torch.tensor([t_len, 2, 4][:b]).to(device)
        sup = [s.to(device) for s in cases.supports_for(filt, adj3d, b)]
        res = []
        for mode in ("batch_major", "time_major"):
            model.zero_grad()
            if mode == "batch_major":
                out = model(x, lengths, sup)                     # model.py:253: the encoder sees x.transpose(0, 1)
            else:
                xt = x.transpose(0, 1).contiguous()
                _, _, last = model.encoder.run(xt, None, sup, lengths=lengths, want_finals=False)
                from eeg_gnn_ssl_amd import ops
                out = ops.cls_head(last.view(b, 19, 64), model.fc.weight, model.fc.bias)
            out.square().sum().backward()
            res.append((out.detach().clone(), [q.grad.clone() for q in model.parameters()]))
        assert torch.equal(res[0][0], res[1][0]), filt
        for a, b_, (nm, _) in zip(res[0][1], res[1][1], model.named_parameters()):
            assert torch.equal(a, b_), (filt, nm, (a - b_).abs().max().item())


# ------------------------------------------------------------------------------------------------
# training-mode dropout (model.py:191,267): masks generated inside the kernels (Philox4x32-10), recomputed in the backward
def philox4x32_10_numpy(counters, seed):
    """Philox4x32-10 (Salmon et al., SC'11) on an array of 64-bit counters under a 64-bit key: (n, 4) uint32 words.
    Reference implementation for the tests (known answer: counter 0, key 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8)."""
    c = np.asarray(counters, dtype=np.uint64)
    c0, c1 = (c & np.uint64(0xffffffff)), (c >> np.uint64(32))
    c2, c3 = np.zeros_like(c0), np.zeros_like(c0)
    k0, k1 = np.uint64(seed & 0xffffffff), np.uint64((seed >> 32) & 0xffffffff)
    m32 = np.uint64(0xffffffff)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        n0, n2 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & m32, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & m32
        c1, c3 = p1 & m32, p0 & m32
        c0, c2 = n0, n2
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m32, (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack([c0, c1, c2, c3], axis=1).astype(np.uint32)


def expected_mask(seed, offset, n, p):
    """the documented function of (seed, offset): element e keeps iff word e%4 of counter offset + e/4 is >= p * 2^32"""
    groups = (n + 3) // 4
    words = philox4x32_10_numpy(np.uint64(offset) + np.arange(groups, dtype=np.uint64), int(seed)).reshape(-1)[:n]
    thr = np.uint32(min(p * 4294967296.0, 4294967295.0))
    return (words >= thr).astype(np.float32) / np.float32(1.0 - p)


def check_dropout_generator(device):
    """known-answer test of the generator + the rng_take_ / dropout_mask operators: the mask is the documented function of
    (seed, offset), rng_take_ hands out consecutive counter ranges, keep rate within 3 sigma."""
    from eeg_gnn_ssl_amd import ops
    assert [f"{w:08x}" for w in philox4x32_10_numpy([0], 0)[0]] == ["6627e8d5", "e169c58d", "bc57ac4c", "9b00dbd8"]
    torch.manual_seed(1234)
    st = ops.make_rng_state(device)
    seed = int(st[0].item())
    assert int(st[1].item()) == 0
    torch.manual_seed(1234)
    assert int(ops.make_rng_state(device)[0].item()) == seed                  # torch.manual_seed governs the seed
    n = 4 * 5000 + 4
    u1 = ops.rng_take(st, n // 4)
    u2 = ops.rng_take(st, 7)
    assert u1.tolist() == [seed, 0] and u2.tolist() == [seed, n // 4] and st.tolist() == [seed, n // 4 + 7]
    for p in (0.5, 0.3):
        m = ops.dropout_mask(u2, n, p).cpu().numpy()
        assert np.array_equal(m, expected_mask(seed, n // 4, n, p)), p
        keep = float((m > 0).mean())
        assert abs(keep - (1 - p)) <= 3 * np.sqrt(p * (1 - p) / n), (p, keep)
    assert float(ops.dropout_mask(u2, 64, 1.0).abs().max()) == 0.0            # p = 1: everything dropped (like torch)
    assert float((ops.dropout_mask(u2, 64, 0.0) - 1).abs().max()) == 0.0      # p = 0: identity


def expected_teacher_flags(seed, offset, seen, decay, t_len):
    """numpy restatement of eeg_dcrnn_teacher_flags: flag t = u_t < k / (k + exp(seen / k)) (utils.py:385-390,
    model.py:194-200), u_t = word t%4 of Philox counter offset + t/4, / 2^32"""
    import math
    groups = (t_len + 3) // 4
    words = philox4x32_10_numpy(np.uint64(offset) + np.arange(groups, dtype=np.uint64), int(seed)).reshape(-1)[:t_len]
    ratio = decay / (decay + math.exp(seen / decay))
    return (words.astype(np.float64) / 4294967296.0 < ratio).astype(np.int32)


def check_teacher_flags(device):
    """known-answer test of the device-side scheduled sampling: the flags are the documented function of the generator pair
    and of the samples-seen counter; generator offset and counter advance on the stream; the flag rate follows the threshold."""
    from eeg_gnn_ssl_amd import ops
    st = torch.tensor([987654321, 5], dtype=torch.int64, device=device)
    seen = torch.tensor([4000], dtype=torch.int64, device=device)
    decay = 3000.0
    f1 = ops.teacher_flags(st, seen, 96, decay, 12)
    f2 = ops.teacher_flags(st, seen, 96, decay, 7)
    assert f1.dtype == torch.int32 and f1.tolist() == expected_teacher_flags(987654321, 5, 4000, decay, 12).tolist()
    assert f2.tolist() == expected_teacher_flags(987654321, 8, 4096, decay, 7).tolist()
    assert st.tolist() == [987654321, 10] and seen.tolist() == [4192]
    # rate: k / (k + exp(n / k)) at n = 0 is k / (k + 1) ~ 1 (always teacher-forced), far out it is ~ 0
    seen0 = torch.zeros(1, dtype=torch.int64, device=device)
    assert sum(ops.teacher_flags(st, seen0, 0, decay, 64).tolist()) >= 63
    far = torch.tensor([60000], dtype=torch.int64, device=device)
    assert sum(ops.teacher_flags(st, far, 0, decay, 64).tolist()) == 0
    mid = torch.tensor([int(decay * np.log(decay))], dtype=torch.int64, device=device)       # ratio = 1/2
    tot = sum(sum(ops.teacher_flags(st, mid, 0, decay, 64).tolist()) for _ in range(16))
    assert abs(tot / 1024.0 - 0.5) < 3 * 0.5 / np.sqrt(1024.0) + 0.01, tot


def check_ssl_device_curriculum(tag, adj3d, device, dropout=0.0, reps=6):
    """DCRNNModel_nextTimePred under curriculum learning with the teacher-forcing flags drawn ON THE DEVICE (batches_seen = a
    device counter tensor): the flags the kernels read are re-derived from the generator pair and handed to the oracle ->
    predictions and every gradient agree; the counter advanced by the increment."""
    from eeg_gnn_ssl_amd import DCRNNModel_nextTimePred, ops, utils
    c = cases.ssl_inputs(tag, adj3d)
    cfg = c["cfg"]
    args = _dropout_args(cfg, dropout)
    args.use_curriculum_learning = True
    args.cl_decay_steps = 50
    model = DCRNNModel_nextTimePred(args, device=device)
    load(model, c["params"], device)
    model.train()
    sup = [t.to(device) for t in c["sup"]]
    x, y = c["x"].to(device), c["y"].to(device)
    b, t_out, n, h = y.shape[0], y.shape[1], 19, cfg.rnn_units
    assert ops.decoder_is_persistent(t_out, b, n, h, cfg.output_dim, 2 * cfg.max_diffusion_step + 1 if cfg.filter_type == "dual_random_walk"
                                     else cfg.max_diffusion_step + 1, cfg.num_rnn_layers)
    seen0 = 196                      # ratio = 50 / (50 + exp(3.92)) ~ 0.5
    seen = torch.tensor([seen0], dtype=torch.int64, device=device)
    model.batches_seen_increment = 24
    model.decoder.set_dropout_seed(20240917, 3)
    seeds = []
    for rep in range(reps):          # several draws: mixed flags must occur and each must match the oracle
        seed, off = model.decoder.dropout_rng_state()
        n_seen = int(seen.item())
        flags = expected_teacher_flags(seed, off, n_seen, 50.0, t_out)
        seeds.append(flags.tolist())
        model.zero_grad()
        pred = model(x, y, sup, batches_seen=seen)
        assert int(seen.item()) == n_seen + 24
        masks = None
        if dropout > 0:
            groups = t_out * b * n * h // 4
            used = torch.tensor([seed, off + (t_out + 3) // 4], dtype=torch.int64, device=device)
            masks = ops.dropout_mask(used, 4 * groups, dropout).view(t_out, b, n, h).cpu()
            assert model.decoder.dropout_rng_state() == (seed, off + (t_out + 3) // 4 + groups)
        else:
            assert model.decoder.dropout_rng_state() == (seed, off + (t_out + 3) // 4)
        loss = utils.compute_regression_loss(y_true=y, y_predicted=pred, standard_scaler=utils.StandardScaler(cases.SSL_MEAN, cases.SSL_STD),
                                             loss_fn="MAE")
        loss.backward()
        uniq, po = {}, {}
        for k, v in c["params"].items():
            if id(v) not in uniq:
                uniq[id(v)] = v.clone().requires_grad_(True)
            po[k] = uniq[id(v)]
        pro = orc.next_time_pred_forward(po, cfg, c["x"], c["y"], c["sup"], teacher_force_mask=[bool(v) for v in flags],
                                         dropout_masks=masks)
        lso = orc.regression_loss(c["y"], pro, cases.SSL_MEAN, cases.SSL_STD, loss_fn="MAE")
        lso.backward()
        assert_close(pred.detach().cpu().numpy(), pro.detach().numpy(), f"device curriculum ssl/{tag}/pred (flags {flags.tolist()})")
        for k, q in model.named_parameters():
            assert_close_scaled(q.grad.cpu().numpy(), po[k].grad.numpy(), f"device curriculum ssl/{tag}/d_{k}")
    assert len({tuple(f) for f in seeds}) > 1 and any(0 < sum(f[:-1]) < t_out - 1 for f in seeds), seeds


def check_device_step_adam(device):
    """eeg_dcrnn_clip_adam_dev (step count and learning rate in device memory, the count advanced by the kernel) walks the same
    parameters as eeg_dcrnn_clip_adam with the host's step / lr arguments, over several steps incl. a learning-rate change."""
    from eeg_gnn_ssl_amd import ops
    g = torch.Generator().manual_seed(5)
    n = 5000
    p0 = torch.randn(n, generator=g)
    grads = [3.0 * torch.randn(n, generator=g) for _ in range(5)]
    res = []
    for dev_side in (False, True):
        p, m, v = p0.clone().to(device), torch.zeros(n, device=device), torch.zeros(n, device=device)
        ws, nrm = torch.zeros(64, device=device), torch.zeros(1, device=device)
        step_dev = torch.zeros(1, dtype=torch.int32, device=device)
        lr_dev = torch.full((1,), 1e-2, device=device)
        norms = []
        for k, g_ in enumerate(grads):
            lr = 1e-2 if k < 3 else 2.5e-3
            gd = g_.clone().to(device)
            if dev_side:
                lr_dev.fill_(lr)
                ops.clip_adam_step_dev(p, gd, m, v, step_dev, lr_dev, (0.9, 0.999), 1e-8, 5e-4, 5.0, 0.5, ws, nrm)
            else:
                ops.clip_adam_step(p, gd, m, v, k + 1, lr, (0.9, 0.999), 1e-8, 5e-4, 5.0, 0.5, ws, nrm)
            norms.append(float(nrm.item()))
        if dev_side:
            assert int(step_dev.item()) == len(grads)
        res.append((p.cpu(), m.cpu(), v.cpu(), norms))
    for a, b_ in zip(res[0][:3], res[1][:3]):
        assert float((a - b_).abs().max()) <= 1e-7 * max(1.0, float(a.abs().max()))
    assert res[0][3] == res[1][3]
    # and against torch.optim.Adam + clip_grad_norm_ (the reference recipe, train.py:222-223,273-275)
    q = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([q], lr=1e-2, weight_decay=5e-4)
    for k, g_ in enumerate(grads):
        for grp in opt.param_groups:
            grp["lr"] = 1e-2 if k < 3 else 2.5e-3
        q.grad = 0.5 * g_.clone()
        torch.nn.utils.clip_grad_norm_([q], 5.0)
        opt.step()
    assert float((q.detach() - res[1][0]).abs().max()) < 2e-6


def _dropout_args(cfg, p):
    a = make_args(cfg)
    a.dropout = p
    return a


def check_dropout_cls_case(tag, golden_dropout, adj3d, device):
    """DCRNNModel_classification in train() mode with dropout 0.5 (README.md:83): the kernel draws its own mask; the mask the
    kernel used is materialised from the generator pair (ops.dropout_mask) and handed to the oracle -> logits and every
    gradient agree; in eval() mode the model is the p = 0 model; the GENUINE reference's training-mode logits (golden,
    closed-form mask) are reproduced when the kernel's mask is replaced by that mask through the same operator."""
    from eeg_gnn_ssl_amd import DCRNNModel_classification, ops
    c = cases.cls_inputs(tag, adj3d)
    p_drop = cases.DROPOUT_P
    model = DCRNNModel_classification(_dropout_args(c["cfg"], p_drop), c["classes"], device=device)
    load(model, c["params"], device)
    model.train()
    sup = [t.to(device) for t in c["sup"]]
    x, seq, y = c["x"].to(device), c["seq"].to(device), c["y"].to(device)
    torch.manual_seed(7)
    logits = model(x, seq, sup)
    st = model._dropout_rng
    b, n, h = x.shape[0], 19, c["cfg"].rnn_units
    used = torch.tensor([int(st[0].item()), int(st[1].item()) - b * n * h // 4], dtype=torch.int64, device=device)
    mask = ops.dropout_mask(used, b * n * h, p_drop).view(b, n, h)
    keep = float((mask > 0).float().mean().item())
    assert abs(keep - (1 - p_drop)) <= 3 * np.sqrt(p_drop * (1 - p_drop) / mask.numel()) + 1e-9, keep
    assert float(mask.max().item()) == 1.0 / (1.0 - p_drop)
    loss = (torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), y) if c["classes"] == 1
            else torch.nn.functional.cross_entropy(logits, y))
    loss.backward()
    po = {k: v.clone().requires_grad_(True) for k, v in c["params"].items()}
    lo = orc.classification_forward(po, c["cfg"], c["x"], c["seq"], c["sup"], dropout_mask=mask.cpu())
    lso = orc.bce_with_logits(lo, c["y"]) if c["classes"] == 1 else orc.cross_entropy(lo, c["y"])
    lso.backward()
    assert_close(logits.detach().cpu().numpy(), lo.detach().numpy(), f"dropout cls/{tag}/logits")
    for k, q in model.named_parameters():
        assert_close_scaled(q.grad.cpu().numpy(), po[k].grad.numpy(), f"dropout cls/{tag}/d_{k}")
    # a second forward draws another mask (the device generator advanced), eval mode drops nothing
    l2 = model(x, seq, sup)
    assert int(model._dropout_rng[1].item()) == int(st[1].item()) and int(st[1].item()) == 2 * (b * n * h // 4)
    assert float((l2 - logits).abs().max().item()) > 0
    model.eval()
    le = model(x, seq, sup)
    le0 = orc.classification_forward(c["params"], c["cfg"], c["x"], c["seq"], c["sup"])
    assert_close(le.detach().cpu().numpy(), le0.detach().numpy(), f"dropout cls/{tag}/eval logits")
    assert int(model._dropout_rng[1].item()) == 2 * (b * n * h // 4)          # eval: the generator does not move


def check_dropout_ssl_case(tag, adj3d, device, teacher=None):
    """DCRNNModel_nextTimePred in train() mode with dropout 0.5 in front of the decoder's projection (model.py:191): a fresh
    in-kernel mask per step; the masks the kernels used (materialised from the generator pair) go to the oracle -> predictions
    and every gradient (incl. the shared decoder cell and dW_p, whose A operand is the dropped rows) agree."""
    from eeg_gnn_ssl_amd import DCRNNModel_nextTimePred, ops
    c = cases.ssl_inputs(tag, adj3d)
    cfg = c["cfg"]
    p_drop = cases.DROPOUT_P
    model = DCRNNModel_nextTimePred(_dropout_args(cfg, p_drop), device=device)
    load(model, c["params"], device)
    model.train()
    sup = [t.to(device) for t in c["sup"]]
    x, y = c["x"].to(device), c["y"].to(device)
    torch.manual_seed(11)
    pred = model(x, y, sup, batches_seen=7)
    st = model.decoder._dropout_rng
    b, t_out, n, h = y.shape[0], y.shape[1], 19, cfg.rnn_units
    groups = t_out * b * n * h // 4
    assert int(st[1].item()) == groups
    used = torch.tensor([int(st[0].item()), 0], dtype=torch.int64, device=device)
    masks = ops.dropout_mask(used, 4 * groups, p_drop).view(t_out, b, n, h)
    keep = float((masks > 0).float().mean().item())
    assert abs(keep - (1 - p_drop)) <= 3 * np.sqrt(p_drop * (1 - p_drop) / masks.numel()) + 1e-9, keep
    for t in range(1, t_out):
        assert not torch.equal(masks[t], masks[0])                             # a fresh draw every step
    from eeg_gnn_ssl_amd import utils
    loss = utils.compute_regression_loss(y_true=y, y_predicted=pred, standard_scaler=utils.StandardScaler(cases.SSL_MEAN, cases.SSL_STD),
                                         loss_fn="MAE")
    loss.backward()
    uniq, po = {}, {}
    for k, v in c["params"].items():
        if id(v) not in uniq:
            uniq[id(v)] = v.clone().requires_grad_(True)
        po[k] = uniq[id(v)]
    pro = orc.next_time_pred_forward(po, cfg, c["x"], c["y"], c["sup"], dropout_masks=masks.cpu())
    lso = orc.regression_loss(c["y"], pro, cases.SSL_MEAN, cases.SSL_STD, loss_fn="MAE")
    lso.backward()
    assert_close(pred.detach().cpu().numpy(), pro.detach().numpy(), f"dropout ssl/{tag}/pred")
    assert abs(loss.item() - lso.item()) < 1e-5
    for k, q in model.named_parameters():
        assert_close_scaled(q.grad.cpu().numpy(), po[k].grad.numpy(), f"dropout ssl/{tag}/d_{k}")
    model.eval()
    pe = model(x, y, sup)
    pe0 = orc.next_time_pred_forward(c["params"], cfg, c["x"], c["y"], c["sup"])
    assert_close(pe.detach().cpu().numpy(), pe0.detach().numpy(), f"dropout ssl/{tag}/eval pred")
