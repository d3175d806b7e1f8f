#!/usr/bin/env python3
"""dev aid: register / LDS / spill counts of the gfx950 kernels of one translation unit.
usage: python tools/regs.py [api|dec|decb|seqs|seq64|seq32|seq16] [name filter ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "eeg_gnn_ssl_amd", "csrc")
unit = sys.argv[1] if len(sys.argv) > 1 else "api"
filt = sys.argv[2:]
if unit == "api":
    src, extra = "api.cpp", []
elif unit in ("spec", "gemmq"):
    src, extra = unit + "_inst.cpp", []
elif unit in ("dec", "decb", "seqs"):
    src, extra = unit + "_inst.cpp", ["-fno-slp-vectorize"]
else:
    src, extra = "seq_inst.cpp", ["-fno-slp-vectorize", f"-DEEG_SEQ_H={unit[3:]}"]
out = f"/tmp/regs_{unit}.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-w",
                       *extra, os.path.join(CSRC, src), "-o", out])
s = open(out).read()
md = s[s.index("amdhsa.kernels"):]
for e in md.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", e).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if filt and not any(f in dem for f in filt):
        continue
    g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", e).group(1))   # noqa: E731
    print(f"{dem[:70]:70s} vgpr {g('vgpr_count'):4d} agpr {int(e.split()[0]):4d} sgpr {g('sgpr_count'):4d} "
          f"spill {g('vgpr_spill_count'):3d} lds {g('group_segment_fixed_size')}")
