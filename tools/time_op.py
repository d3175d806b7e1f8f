"""dev aid (GPU box): HIP-event timing of the pre-processing operators at cfg2 size.
usage: python tools/time_op.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_gnn_ssl_amd import _lib, ops
lib = _lib.get_lib()
dev = "cuda"
b, n, t, w = 256, 19, 60, 200
raw = torch.randn(b, n, t * w, device=dev) * 20
for _ in range(2):
    fr, fs = ops.fft_features(raw, w, 3.924, 1.56)
    sup = ops.correlation_supports(fr)
torch.cuda.synchronize()
lib.query("eeg_dcrnn_prof_enable", 1)
for _ in range(10):
    fr, fs = ops.fft_features(raw, w, 3.924, 1.56)
    sup = ops.correlation_supports(fr)
torch.cuda.synchronize()
lib.query("eeg_dcrnn_prof_enable", 0)
buf = ctypes.create_string_buffer(1 << 16)
lib.call("eeg_dcrnn_prof_report", buf, len(buf))
agg = {}                                          # report lines are "role count total_ms symbol" (one per role AND kernel symbol; symbols contain blanks)
for line in buf.value.decode().strip().splitlines():
    name, cnt, ms = line.split(None, 3)[:3]
    c, m = agg.get(name, (0, 0.0))
    agg[name] = (c + int(cnt), m + float(ms))
for name, (cnt, ms) in agg.items():
    print(f"{name:16s} {ms / cnt * 1e3:9.1f} us / launch")
print("raw signals", raw.numel() * 4 / 1e6, "MB; features", fr.numel() * 4 / 1e6, "MB")
