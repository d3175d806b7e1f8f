"""dev aid: why does the streamed-input pass slow down when a (world size 1) RCCL process group issues the all-reduce?
host-side timestamps of fetch / replay / all-reduce per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from eeg_gnn_ssl_amd import DCRNNModel_classification
from eeg_gnn_ssl_amd.train_step import TrainStep
mode = sys.argv[1]           # none | pg | reduce
dev = torch.device("cuda", 0)
if mode != "none":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1)
task, filt, t_len, batch, classes = bench.WORKLOADS["cfg2"]
torch.manual_seed(1)
model = DCRNNModel_classification(bench.make_args(filt), classes, device=dev).to(dev).train()
st = TrainStep(model, task=task, always_reduce=(mode == "reduce"))
hx, hy, hlen, hsup = bench.synthetic_batch(task, filt, t_len, batch, classes, seed=1)
x, y, lengths, sup = hx.to(dev), hy.to(dev), hlen.to(dev), [s.to(dev) for s in hsup]
order = sys.argv[2] if len(sys.argv) > 2 else "A"
st.capture(x, y, lengths, sup, slot=0)
if order == "B":                 # like bench.py: many reduced steps first, the second input set is captured afterwards
    for _ in range(20):
        st.replay_step(0)
    torch.cuda.synchronize()
x2, y2 = x.clone(), y.clone()
st.capture(x2, y2, lengths, sup, slot=1)
sets = [(x, y), (x2, y2)]
pin = [hx.pin_memory(), hy.pin_memory()]
side = torch.cuda.Stream()
landed = [torch.cuda.Event() for _ in sets]; done = [torch.cuda.Event() for _ in sets]
for e in done: e.record()
ev = []
def fetch(i):
    with torch.cuda.stream(side):
        side.wait_event(done[i])
        a = torch.cuda.Event(enable_timing=True); a.record(side)
        sets[i][0].copy_(pin[0], non_blocking=True); sets[i][1].copy_(pin[1], non_blocking=True); landed[i].record(side)
        b = torch.cuda.Event(enable_timing=True); b.record(side)
        ev.append(("copy", a, b))
fetch(0)
for _ in range(3):
    st.replay_step(0)
torch.cuda.synchronize()
base = torch.cuda.Event(enable_timing=True); base.record()
ev.clear()
rows = []
t_all = time.perf_counter()
for k in range(12):
    i = k % 2
    t0 = time.perf_counter(); fetch((i + 1) % 2)
    t1 = time.perf_counter(); torch.cuda.current_stream().wait_event(landed[i])
    a = torch.cuda.Event(enable_timing=True); a.record(); st._graphs[i][0].replay()
    b = torch.cuda.Event(enable_timing=True); b.record()
    t2 = time.perf_counter(); st.reduce_and_update(); done[i].record()
    c = torch.cuda.Event(enable_timing=True); c.record(); ev.append(("graph", a, b)); ev.append(("tail", b, c))
    t3 = time.perf_counter(); rows.append((t1 - t0, t2 - t1, t3 - t2))
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print(mode, f"{tot / 12 * 1e3:.3f} ms/step; host ms per step (fetch, replay, reduce+update):", " | ".join(f"{a*1e3:.2f} {b*1e3:.2f} {c*1e3:.2f}" for a, b, c in rows[4:8]))

for name, a, b in ev[-16:]:
    print(f"  {name:6s} {base.elapsed_time(a):8.3f} -> {base.elapsed_time(b):8.3f} ms")
if mode != "none":
    dist.destroy_process_group()
