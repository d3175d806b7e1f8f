#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, build container only (needs /root/reference; never runs on the GPU box): randomized DIFFERENTIAL check of the
oracle (oracle/dcrnn_oracle.py) against the GENUINE reference classes imported from /root/reference, beyond the committed goldens --
classification and SSL models over filter types, 0..3 diffusion steps, 5 / 19 / 20 nodes, 16 / 64 units, 1..4 layers (shared decoder
cell), tanh / relu, ragged lengths: outputs must be bit-equal on the CPU (same torch ops in the same order), every parameter gradient
within 2e-6 of the tensor's largest entry (autograd accumulation order).  Import recipe of SURVEY.md 8(c): h5py / pyedflib stubs and
`Tensor.cuda` as identity -- which is why this runs as its own process.
usage: python tests/golden/oracle_vs_reference.py [--cases 24] [--seed 7]"""
import argparse
import os
import random
import sys
import types
import warnings

REF = "/root/reference"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=24)
    ap.add_argument("--seed", type=int, default=7)
    a = ap.parse_args()
    if not os.path.isdir(REF):
        print("reference tree not present: nothing to compare")
        return 0
    warnings.simplefilter("ignore")
    for m in ("h5py", "pyedflib"):
        sys.modules[m] = types.ModuleType(m)
    sys.path.insert(0, REF)
    import torch
    torch.Tensor.cuda = lambda self, *args, **kw: self           # model.py:336 hard-codes .cuda()
    from model.model import DCRNNModel_classification, DCRNNModel_nextTimePred
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import dcrnn_oracle as orc
    rng = random.Random(a.seed)
    worst_out, worst_grad = 0.0, 0.0
    for case in range(a.cases):
        task = "cls" if case % 2 == 0 else "ssl"
        filt = rng.choice(["laplacian", "random_walk", "dual_random_walk"])
        k, n, h = rng.choice([0, 1, 2, 3]), rng.choice([5, 19, 20]), rng.choice([16, 64])
        din, layers, act = rng.choice([4, 12]), rng.choice([1, 2, 3, 4]), rng.choice(["tanh", "relu"])
        b, t_len, t_out, classes = rng.choice([1, 3]), rng.choice([1, 4]), rng.choice([1, 3]), rng.choice([1, 4])
        args = types.SimpleNamespace(num_nodes=n, num_rnn_layers=layers, rnn_units=h, input_dim=din, output_dim=din,
                                     max_diffusion_step=k, dcgru_activation=act, filter_type=filt, dropout=0.0, cl_decay_steps=3000,
                                     use_curriculum_learning=False)
        torch.manual_seed(case)
        g = torch.Generator().manual_seed(1000 + case)
        sup = [torch.rand(b, n, n, generator=g) for _ in range(2 if filt == "dual_random_walk" else 1)]
        x = torch.randn(b, t_len, n, din, generator=g)
        cfg = orc.DCRNNConfig(num_nodes=n, filter_type=filt, input_dim=din, output_dim=din, rnn_units=h, num_rnn_layers=layers,
                              num_classes=classes, max_diffusion_step=k, dcgru_activation=act)
        ref = (DCRNNModel_classification(args, classes, device="cpu") if task == "cls" else DCRNNModel_nextTimePred(args, device="cpu")).eval()
        uniq, params = {}, {}
        for name, v in ref.state_dict().items():                 # shared tensors (decoder cell of layers >= 1) stay shared
            key = v.data_ptr()
            if key not in uniq:
                uniq[key] = v.detach().clone().requires_grad_(True)
            params[name] = uniq[key]
        if task == "cls":
            lens = torch.tensor([t_len] + [max(1, t_len - 1)] * (b - 1))
            w = torch.randn(b, classes, generator=g)
            out_r = ref(x, lens, sup)
            out_o = orc.classification_forward(params, cfg, x, lens, sup)
        else:
            y = torch.randn(b, t_out, n, din, generator=g)
            w = y
            out_r = ref(x, y, sup)
            out_o = orc.next_time_pred_forward(params, cfg, x, y, sup)
        (out_r * w).sum().backward()
        (out_o * w).sum().backward()
        d_out = float((out_r - out_o).abs().max())
        d_grad = max(float((p.grad - params[name].grad).abs().max() / max(1e-6, float(p.grad.abs().max()))) for name, p in ref.named_parameters())
        worst_out, worst_grad = max(worst_out, d_out), max(worst_grad, d_grad)
        print(f"{task} {filt} K={k} N={n} H={h} D={din} L={layers} {act} B={b} T={t_len}"
              + (f" C={classes}" if task == "cls" else f" T_out={t_out}") + f": outputs differ by {d_out:.1e}, gradients by {d_grad:.1e} (rel)")
        if d_out != 0.0 or d_grad > 2e-6:
            print("MISMATCH")
            return 1
    print(f"oracle vs genuine reference: {a.cases} random cases, outputs bit-equal, worst gradient difference {worst_grad:.1e} (rel)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
