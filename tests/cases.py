"""Rebuild the closed-form inputs of every golden case (must mirror tests/golden/make_golden.py).

Used by the oracle-vs-golden tests (CPU) and by the HIP-vs-golden parity tests (GPU), so both
check against exactly the inputs the genuine reference was run on."""
import numpy as np
import torch

from closed_form import cf, cf_adjacency, cf_dropout_mask, cf_params
from oracle import dcrnn_oracle as orc

N = 19


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def lap_supports(adj3d, b, batched=True):
    s = orc.compute_supports(adj3d, "laplacian")[0]
    return [s.unsqueeze(0).repeat(b, 1, 1)] if batched else [s]


def dual_supports(b, phase0=0.3):
    s1, s2 = [], []
    for i in range(b):
        a = orc.keep_topk(cf_adjacency(N, phase=phase0 + 1.7 * i), top_k=3, directed=True)
        s = orc.compute_supports(a, "dual_random_walk")
        s1.append(s[0])
        s2.append(s[1])
    return [torch.stack(s1), torch.stack(s2)]


def supports_for(filt, adj3d, b, batched=True):
    if filt == "random_walk":
        return [dual_supports(b)[0]]
    return dual_supports(b) if filt == "dual_random_walk" else lap_supports(adj3d, b, batched)


DCONV_CASES = {
    "lap_small": ("laplacian", 8, 16, 32, 3, True),
    "lap_small_unbatched": ("laplacian", 8, 16, 32, 3, False),
    "dual_small": ("dual_random_walk", 8, 16, 32, 3, True),
    "lap_default": ("laplacian", 100, 64, 128, 2, True),
    "dual_default": ("dual_random_walk", 100, 64, 128, 2, True),
}


def dconv_inputs(tag, adj3d):
    filt, din, h, o, b, batched = DCONV_CASES[tag]
    m = 5 if filt == "dual_random_walk" else 3
    p = cf_params({"weight": ((din + h) * m, o), "biases": (o,)}, base_phase=0.5)
    sup = supports_for(filt, adj3d, b, batched)
    x = T(cf((b, N * din), scale=1.0, freq=0.371, phase=0.1))
    s = T(cf((b, N * h), scale=0.8, freq=0.533, phase=0.7))
    return dict(filt=filt, din=din, h=h, o=o, b=b, sup=sup, x=x, s=s,
                weight=T(p["weight"]), biases=T(p["biases"]))


CELL_CASES = {
    "lap_small": ("laplacian", 8, 16, 3, "tanh", True),
    "dual_small": ("dual_random_walk", 8, 16, 3, "tanh", True),
    "lap_small_relu": ("laplacian", 8, 16, 3, "relu", True),
    "lap_default": ("laplacian", 100, 64, 2, "tanh", False),
    "dual_default": ("dual_random_walk", 100, 64, 2, "tanh", False),
    "lap_l1_default": ("laplacian", 64, 64, 2, "tanh", False),
}


# other diffusion orders / filter types (tests/golden/make_golden_k.py): (filter, din, h, b, act, K)
CELL_K_CASES = {
    "lap_k1": ("laplacian", 8, 16, 3, "tanh", 1),
    "lap_k3": ("laplacian", 8, 16, 3, "tanh", 3),
    "dual_k1": ("dual_random_walk", 8, 16, 3, "tanh", 1),
    "dual_k3": ("dual_random_walk", 8, 16, 3, "relu", 3),
    "rw_k2": ("random_walk", 12, 32, 2, "tanh", 2),
    # max_diffusion_step = 0 (tests/golden/make_golden_k0.py): one "hop matrix" = the identity, no graph mixing
    "lap_k0": ("laplacian", 8, 16, 3, "tanh", 0),
    "dual_k0_h64": ("dual_random_walk", 100, 64, 2, "tanh", 0),
}


def cell_shapes(filt, din, h, k=2):
    m = (2 if filt == "dual_random_walk" else 1) * k + 1
    rows = (din + h) * m
    return {"dconv_gate.weight": (rows, 2 * h), "dconv_gate.biases": (2 * h,),
            "dconv_candidate.weight": (rows, h), "dconv_candidate.biases": (h,)}


def cell_inputs(tag, adj3d):
    if tag in CELL_K_CASES:
        (filt, din, h, b, act, kk), full = CELL_K_CASES[tag], True
    else:
        (filt, din, h, b, act, full), kk = CELL_CASES[tag], 2
    p = {k: T(v) for k, v in cf_params(cell_shapes(filt, din, h, kk), base_phase=1.1).items()}
    sup = supports_for(filt, adj3d, b)
    x = T(cf((b, N * din), scale=1.0, freq=0.371, phase=0.1))
    s = T(cf((b, N * h), scale=0.8, freq=0.533, phase=0.7))
    wout = T(cf((b, N * h), scale=1.0, freq=0.291, phase=0.4))
    return dict(filt=filt, din=din, h=h, b=b, act=act, full=full, params=p, sup=sup, x=x, s=s, wout=wout, k=kk)


CLS_CASES = {
    # tag: (filter, din, h, layers, classes, b, t, lengths, full)
    "lap_small_bce": ("laplacian", 8, 16, 2, 1, 3, 5, None, True),
    "lap_small_ce_varlen": ("laplacian", 8, 16, 2, 4, 4, 6, [6, 3, 5, 1], True),
    "dual_small_bce": ("dual_random_walk", 8, 16, 2, 1, 3, 5, None, True),
    "lap_default_bce": ("laplacian", 100, 64, 2, 1, 4, 12, None, False),
    "dual_default_bce": ("dual_random_walk", 100, 64, 2, 1, 3, 6, None, False),
    "lap_default_ce_varlen": ("laplacian", 100, 64, 2, 4, 3, 8, [8, 5, 2], False),
}


def cls_inputs(tag, adj3d):
    filt, din, h, layers, classes, b, t, lengths, full = CLS_CASES[tag]
    cfg = orc.DCRNNConfig(filter_type=filt, input_dim=din, rnn_units=h, num_rnn_layers=layers,
                          num_classes=classes)
    p = {k: T(v) for k, v in cf_params(orc.param_shapes(cfg, "classification"), base_phase=2.3).items()}
    sup = supports_for(filt, adj3d, b)
    x = T(cf((b, t, N, din), scale=1.0, freq=0.4177, phase=0.9))
    if lengths is None:
        lengths = [t] * b
    else:
        for i, ln in enumerate(lengths):
            x[i, ln:] = 0
    seq = torch.tensor(lengths, dtype=torch.int64)
    if classes == 1:
        y = T((cf((b,), scale=1.0, freq=2.1, phase=0.3) > 0).astype(np.float32))
    else:
        y = torch.tensor([(3 * i + 1) % classes for i in range(b)], dtype=torch.int64)
    return dict(cfg=cfg, params=p, sup=sup, x=x, seq=seq, y=y, full=full, classes=classes)


SSL_CASES = {
    # tag: (filter, din, h, layers, b, t_in, t_out, full)
    "lap_small": ("laplacian", 8, 16, 2, 3, 4, 3, True),
    "dual_small_L3": ("dual_random_walk", 8, 16, 3, 2, 4, 3, True),
    "dual_default": ("dual_random_walk", 100, 64, 2, 2, 5, 3, False),
}
SSL_MEAN, SSL_STD = 3.924, 1.560


def ssl_inputs(tag, adj3d):
    filt, din, h, layers, b, t_in, t_out, full = SSL_CASES[tag]
    cfg = orc.DCRNNConfig(filter_type=filt, input_dim=din, output_dim=din, rnn_units=h, num_rnn_layers=layers)
    raw = cf_params(orc.param_shapes(cfg, "ssl"), base_phase=3.7)
    for l in range(2, layers):      # Q6: one shared decoder cell for layers >= 1
        for k in list(raw):
            if k.startswith(f"decoder.decoding_cells.{l}."):
                raw[k] = raw[k.replace(f"decoding_cells.{l}.", "decoding_cells.1.")]
    p = {}
    for k, v in raw.items():
        src = k
        if k.startswith("decoder.decoding_cells."):
            l = int(k.split(".")[2])
            if l >= 2:
                src = k.replace(f"decoding_cells.{l}.", "decoding_cells.1.")
        p[k] = p[src] if (src != k and src in p) else T(v)
    # make sure shared entries are literally the same tensor object
    for k in list(p):
        if k.startswith("decoder.decoding_cells."):
            l = int(k.split(".")[2])
            if l >= 2:
                p[k] = p[k.replace(f"decoding_cells.{l}.", "decoding_cells.1.")]
    sup = supports_for(filt, adj3d, b)
    x = T(cf((b, t_in, N, din), scale=1.0, freq=0.4177, phase=0.9))
    y = T(cf((b, t_out, N, din), scale=1.0, freq=0.3319, phase=1.9))
    y[0, 0, 0, :3] = 0.0
    return dict(cfg=cfg, params=p, sup=sup, x=x, y=y, full=full)


def train_inputs(adj3d):
    """the closed-form training task of tests/golden/make_golden_train.py"""
    from closed_form import train_task
    cfg = orc.DCRNNConfig(filter_type="laplacian", input_dim=100, rnn_units=64, num_rnn_layers=2, num_classes=1)
    x, y = (T(a) for a in train_task(32, 12))
    return dict(cfg=cfg, x=x, y=y, seq=torch.full((32,), 12, dtype=torch.long), sup=lap_supports(adj3d, 32),
                base_phase=4.1)


def ssl_train_inputs(golden_train):
    """the closed-form SSL training task of tests/golden/make_golden_train.py (shared decoder cell, 3 layers)"""
    steps, lr, wd, clip, b, t_in, t_out, mean, std = (float(v) for v in golden_train["ssl_train/hparams"])
    b, t_in, t_out = int(b), int(t_in), int(t_out)
    cfg = orc.DCRNNConfig(filter_type="dual_random_walk", input_dim=20, output_dim=20, rnn_units=32, num_rnn_layers=3)
    raw = cf_params(orc.param_shapes(cfg, "ssl"), base_phase=5.3)
    p = {}
    for k, v in raw.items():                              # Q6: decoding_cells.2 IS decoding_cells.1
        src = k.replace("decoding_cells.2.", "decoding_cells.1.")
        p[k] = p[src] if (src != k and src in p) else T(raw[src])
    for k in list(p):
        if "decoding_cells.2." in k:
            p[k] = p[k.replace("decoding_cells.2.", "decoding_cells.1.")]
    x = T(cf((b, t_in, N, 20), scale=1.0, freq=0.4177, phase=0.9))
    y = T(cf((b, t_out, N, 20), scale=1.0, freq=0.3319, phase=1.9))
    y[0, 0, 0, :3] = -mean / std
    return dict(cfg=cfg, params=p, x=x, y=y, sup=dual_supports(b), steps=int(steps), lr=lr, wd=wd, clip=clip,
                mean=mean, std=std)


# ---- training-mode dropout (tests/golden/make_golden_dropout.py: the genuine reference run with closed-form masks) ----------
DROPOUT_P = 0.5
DROPOUT_CLS_TAGS = ("lap_small_ce_varlen", "dual_small_bce", "lap_default_ce_varlen")      # tags of CLS_CASES
DROPOUT_SSL_TAGS = ("lap_small", "dual_small_L3", "dual_default")                          # tags of SSL_CASES


def dropout_cls_mask(tag):
    """mask x 1/(1-p) the reference's classification head was given in the golden run: (B,N,H)"""
    _, _, h, _, _, b, _, _, _ = CLS_CASES[tag]
    return T(cf_dropout_mask((b, N, h), DROPOUT_P, 0.9))


def dropout_ssl_masks(tag):
    """one mask per decoder step, in call order: (T_out,B,N,H)"""
    _, _, h, _, b, _, t_out, _ = SSL_CASES[tag]
    return torch.stack([T(cf_dropout_mask((b, N, h), DROPOUT_P, 1.3 + 0.37 * t)) for t in range(t_out)])
