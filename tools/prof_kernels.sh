#!/bin/bash
# rocprofv3 kernel trace of a short bench run; prints per-kernel-name (incl. template args) stats.  args: extra bench flags
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o ktrace -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-prof "$@" > "$OLDPWD/gpurun_out/rocprof.log" 2>&1 )
python - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/prof/ktrace_kernel_trace.csv')))
agg=collections.defaultdict(list)
for r in rows:
    name=r['Kernel_Name']
    key=(name.split('(')[0][:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''), r.get('Workgroup_Size_X', r.get('Workgroup_Size','')))
    agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=sum(sum(v) for v in agg.values())
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:24]:
    print(f"{sum(v)/7:9.1f} us/step  n={len(v):3d} avg={sum(v)/len(v):8.1f} min={min(v):8.1f}  grid={k[1]:>8} wg={k[2]:>4}  {k[0]}")
print(f"total {tot/7:.1f} us/step")
PY
rm -f gpurun_out/prof/ktrace_kernel_trace.csv
