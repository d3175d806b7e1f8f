#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) result as a per-kernel table: tools/rocpd_stats.py x.db [steps]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                       "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x/workgroup_x) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"# total kernel time {tot:.3f} ms over {steps:g} steps -> {tot/steps:.3f} ms/step")
print("# total_ms  pct  calls  avg_us  min_us  max_us  vgpr agpr lds_bytes wgs  name")
for r in rows:
    print(f"{r[2]:9.3f} {100*r[2]/tot:5.1f} {r[1]:6d} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f} {r[6]:4d} {r[7]:4d} {r[8]:7d} {r[9]:6d}  {r[0][:110]}")
