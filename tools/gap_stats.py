#!/usr/bin/env python3
"""Idle time between consecutive kernels of the replayed step, from a rocprofv3 kernel trace (CSV).
usage: python tools/gap_stats.py <kernel_trace.csv> [steps_to_skip]
Prints, for the steady-state part of the trace: GPU-busy time and idle gaps per step, the gap in front of every kernel
symbol (mean), and the share of the step spent in kernels shorter than 12 us."""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
# a step is cut at a kernel that runs exactly once per step: the weight packs of the encoder (the window [marker k, marker k+1)
# holds the launches of one step in steady state)
starts = [i for i, r in enumerate(rows) if "pack_cells_kernel" in r[2]]
if len(starts) < 3:      # (workloads that pack cell by cell: the gradient-norm partials of the update, once per step without --graph-update's extra loops)
    starts = [i for i, r in enumerate(rows) if "sqnorm_partial" in r[2]]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 15
starts = starts[skip:]
if len(starts) < 3:
    sys.exit("not enough steps in the trace")
lo, hi = starts[0], starts[-1]
steps = len(starts) - 1
seg = rows[lo:hi]
busy = sum(e - s for s, e, _ in seg)
span = rows[hi][0] - rows[lo][0]
gap_by = collections.defaultdict(list)
small = 0
for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
    gap_by[n1].append(max(0, s1 - e0))
for s, e, n in seg:
    if e - s < 12000:
        small += e - s
print(f"{steps} steps: span {span / steps / 1e3:.1f} us/step, kernels {busy / steps / 1e3:.1f} us/step, idle {(span - busy) / steps / 1e3:.1f} us/step "
      f"({100.0 * (span - busy) / span:.1f} %), kernels per step {len(seg) / steps:.1f}, time in kernels < 12 us: {small / steps / 1e3:.1f} us/step")
print("gap in front of (mean us, per step count):")
for n, g in sorted(gap_by.items(), key=lambda kv: -sum(kv[1])):
    print(f"   {n:60s} {sum(g) / len(g) / 1e3:7.2f} x {len(g) / steps:4.1f} = {sum(g) / steps / 1e3:7.1f} us/step")
