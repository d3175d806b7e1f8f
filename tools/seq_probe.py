#!/usr/bin/env python3
"""Development aid (GPU box): per-phase shader-clock cycles of the recurrent kernels at cfg2 size.
usage: python tools/seq_probe.py [workload] [dev-library path]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from eeg_gnn_ssl_amd import DCRNNModel_classification, _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
task, filt, t_len, batch, classes = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
x, y, lengths, sup = bench.synthetic_batch(task, filt, t_len, batch, classes, seed=123)
model = DCRNNModel_classification(bench.make_args(filt), classes, device=dev).to(dev).train()
x, y, lengths, sup = x.to(dev), y.to(dev), lengths.to(dev), [s.to(dev) for s in sup]
if os.environ.get("EEG_PROBE_SHARED", "1") != "0":       # the distance graph in its shared form: the spectral kernels (as TrainStep runs them)
    from eeg_gnn_ssl_amd import ops
    sup = ops.collapse_shared_supports(sup)
_lib._LIB = _lib.EegDcrnnLib(os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else _lib.DEV_LIB_PATH, strict=False)   # cycle probe: dev build only
lib = _lib.get_lib()


def run():
    model.zero_grad()
    lg = model(x, lengths, sup)
    loss = (torch.nn.functional.binary_cross_entropy_with_logits(lg.view(-1), y) if classes == 1
            else torch.nn.functional.cross_entropy(lg, y))
    loss.backward()
    torch.cuda.synchronize()


run()
probe = torch.zeros(batch * 4 * 32, dtype=torch.int64, device=dev)
lib.query("eeg_dcrnn_set_seq_probe", ctypes.c_void_p(probe.data_ptr()))
run()          # last launches of each direction (layer 0 bwd, layer 1 fwd) leave their counters
lib.query("eeg_dcrnn_set_seq_probe", None)
p = probe.view(batch, 4, 32).double().cpu()
names_f = ["loop top + barrier(1)", "gate GEMM", "gate epilogue", "barrier(2)", "cand GEMM (16-node part)", "cand epilogue", "barrier(3) + diffuse(h)", "diffuse(rh) issue"]   # chain waves of seq_fwd2_kernel
names_b = ["E1", "barrier(1)", "GEMM1", "epi1", "adj diffuse dG+bar", "GEMM2", "operand copy + prefetch issue", "adj diffuse dC issue"]
for title, off, names in (("seq_fwd", 0, names_f), ("seq_bwd", 8, names_b)):
    tot = p[:, :, off:off + 8].sum(-1).mean().item() / t_len
    print(f"{title}: {tot:9.0f} cycles/step/wave (mean over {batch} WGs x 4 waves)")
    for k, nm in enumerate(names):
        v = p[:, :, off + k] / t_len
        print(f"   {nm:22s} mean {v.mean().item():8.0f}  min {v.min().item():8.0f}  max {v.max().item():8.0f}   per-wave means "
              + " ".join(f"{v[:, w].mean().item():7.0f}" for w in range(4)))

# chip-wide 100 MHz stamps of the chain waves: kernel entry / first step / behind the last step / kernel exit
for title, off in (("seq_fwd", 16), ("seq_bwd", 20)):
    r = p[:, :, off:off + 4]
    if float(r[:, :, 1].min()) <= 0:
        continue
    t0 = r[:, :, 0].min()
    pro = (r[:, :, 1] - r[:, :, 0]) / 100.0
    loop = (r[:, :, 2] - r[:, :, 1]) / 100.0
    epi = (r[:, :, 3] - r[:, :, 2]) / 100.0
    cyc = p[:, :, (0 if off == 16 else 8):(8 if off == 16 else 16)].sum(-1)
    print(f"{title}: prologue {pro.mean().item():6.1f} us (max {pro.max().item():.1f}), time loop {loop.mean().item():6.1f} us, epilogue {epi.mean().item():5.1f} us; "
          f"first entry -> last exit {(r[:, :, 3].max() - t0).item() / 100.0:6.1f} us, entry spread {((r[:, :, 0].max() - t0) / 100.0).item():.1f} us; "
          f"shader clock inside the loop {(cyc / (loop * 1e-6)).mean().item() / 1e6:7.1f} MHz")
