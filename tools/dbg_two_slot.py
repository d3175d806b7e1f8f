import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from eeg_gnn_ssl_amd import DCRNNModel_classification
from eeg_gnn_ssl_amd.train_step import TrainStep
DEV = "cuda"
task, filt, classes = "detection", "laplacian", 1
batches = [bench.synthetic_batch(task, filt, 9, 6, classes, seed=30 + i) for i in range(5)]
sup = [s.to(DEV) for s in batches[0][3]]
ld = batches[0][2].to(DEV)
def run(mode):
    torch.manual_seed(1)
    model = DCRNNModel_classification(bench.make_args(filt), classes, device=DEV).to(DEV).train()
    st = TrainStep(model, task=task, lr=1e-3)
    traj = []
    if mode == "eager":
        for x, y, _, _ in batches:
            st.step(x.to(DEV), y.to(DEV), ld, sup); torch.cuda.synchronize(); traj.append(st.fp.flat.clone())
    else:
        nslot = 2 if mode == "two" else 1
        bufs = [(torch.zeros_like(batches[0][0], device=DEV), torch.zeros_like(batches[0][1], device=DEV)) for _ in range(nslot)]
        for s in range(nslot):
            st.capture(bufs[s][0], bufs[s][1], ld, sup, slot=s)
        torch.cuda.synchronize()
        for k, (x, y, _, _) in enumerate(batches):
            s = k % nslot
            bufs[s][0].copy_(x.to(DEV)); bufs[s][1].copy_(y.to(DEV)); torch.cuda.synchronize()
            st.replay_step(s); torch.cuda.synchronize(); traj.append(st.fp.flat.clone())
    return traj
e, o, t = run("eager"), run("one"), run("two")
for k in range(5):
    print(k, "one-slot vs eager", (o[k] - e[k]).abs().max().item(), "two-slot vs eager", (t[k] - e[k]).abs().max().item())
