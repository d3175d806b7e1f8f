"""dev aid: parameter gradients of one cfg2-size step under different tuning-knob sets, compared with the first set."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from eeg_gnn_ssl_amd import DCRNNModel_classification, _lib, ops
from eeg_gnn_ssl_amd.train_step import TrainStep
dev = "cuda"
wl = sys.argv[1]
sets = sys.argv[2:]
task, filt, t_len, batch, classes = bench.WORKLOADS[wl]
x, y, lengths, sup = bench.synthetic_batch(task, filt, t_len, batch, classes, seed=123)
x, y, lengths, sup = x.to(dev), y.to(dev), lengths.to(dev), [s.to(dev) for s in sup]
_lib._LIB = _lib.EegDcrnnLib(_lib.DEV_LIB_PATH)      # the tuning knobs exist in the dev build only
lib = _lib.get_lib()
ref = None
for st in sets:
    for k in range(16):
        lib.call("eeg_dcrnn_set_tuning", k, 0)
    if st != "-":
        for kv in st.split(","):
            k, v = kv.split("="); lib.call("eeg_dcrnn_set_tuning", int(k), int(v))
    torch.manual_seed(123)
    model = DCRNNModel_classification(bench.make_args(filt), classes, device=dev).to(dev).train()
    ts = TrainStep(model, task=task)
    loss = ts.forward_backward(x, y, lengths, sup)
    g = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    if ref is None:
        ref = g; print(st, "loss", loss.item()); continue
    worst = max(((g[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-12)).item() for k in g)
    print(st, "loss", loss.item(), "worst rel grad diff vs first:", f"{worst:.3e}",
          {k: f"{((g[k]-ref[k]).abs().max()/ref[k].abs().max().clamp_min(1e-12)).item():.1e}" for k in g if ((g[k]-ref[k]).abs().max()/ref[k].abs().max().clamp_min(1e-12)).item() > 1e-4})
