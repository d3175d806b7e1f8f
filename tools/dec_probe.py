#!/usr/bin/env python3
"""Development aid (GPU box): per-phase shader-clock cycles of dec_fwd_persist_kernel at cfg5 size (SSL model, B=512, 12 decoder steps).
Needs a dev library built with -DEEG_DEC_PROBE (make dev XFLAGS=-DEEG_DEC_PROBE ...).  usage: python tools/dec_probe.py <dev-library> [layers] [bwd]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eeg_gnn_ssl_amd import DCRNNModel_nextTimePred, _lib  # noqa: E402

layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
task, filt, t_len, batch, classes = bench.WORKLOADS["cfg5"]
dev = torch.device("cuda", 0)
x, y, lengths, sup = bench.synthetic_batch(task, filt, t_len, batch, classes, seed=123)
_lib._LIB = _lib.EegDcrnnLib(os.path.abspath(sys.argv[1]), strict=False)
lib = _lib.get_lib()
model = DCRNNModel_nextTimePred(bench.make_args(filt, 0.0, layers), device=dev).to(dev).train()
x, y, sup = x.to(dev), y.to(dev), [s.to(dev) for s in sup]
t_out = bench.T_OUT


bwd = len(sys.argv) > 3 and sys.argv[3] == "bwd"


def run():
    if bwd:
        model.zero_grad()
        model(x, y, sup, None).square().mean().backward()
    else:
        with torch.no_grad():
            model(x, y, sup, None)
    torch.cuda.synchronize()


run()
nwg = 256
probe = torch.zeros(nwg * 4 * 32, dtype=torch.int64, device=dev)
lib.query("eeg_dcrnn_set_seq_probe", ctypes.c_void_p(probe.data_ptr()))
run()
lib.query("eeg_dcrnn_set_seq_probe", None)
p = probe.view(nwg, 4, 32).double().cpu()
# (bwd: the backward kernel ran last and uses the same slots: its counters are the ones left)
names_b = ["output-gradient tile", "projection transpose", "blend backward + node mix + barrier", "GEMM1 (+ GEMM2 prefetch)",
           "GEMM1 epilogue + node mixes + barrier", "GEMM2", "dX hand-over + barrier", "clip set-up / bias sums"]
names = ["input node mix + planes", "x-part GEMM (+ gate prefetch)", "barrier", "gate GEMM", "gate epilogue + node mix + barrier",
         "candidate GEMM", "cand epilogue + node mix + barrier", "projection + feedback + barrier"]
per = (batch / nwg) * t_out       # (clip, step) pairs per workgroup
if bwd:
    names = names_b
tot = p[:, :, :16].sum(-1).mean().item() / per
print(f"{'dec_bwd_persist' if bwd else 'dec_fwd_persist'}: {tot:9.0f} cycles per (clip, step) and wave, {layers} layers")
for off, title in ((0, "layer 0 / per-step phases"), (8, "layers above")):
    for k, nm in enumerate(names):
        v = p[:, :, off + k] / per
        if float(v.abs().max()) == 0:
            continue
        print(f"   [{title:26s}] {nm:36s} mean {v.mean().item():8.0f}  min {v.min().item():8.0f}  max {v.max().item():8.0f}   per-wave means "
              + " ".join(f"{v[:, w].mean().item():7.0f}" for w in range(4)))
