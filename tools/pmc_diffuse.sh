#!/bin/bash
# HBM traffic of the diffusion kernel from PMC counters (separate passes for FETCH_SIZE / WRITE_SIZE,
# counters only: no trace domains).  Prints per-dispatch averages for diffuse_fwd_stream_kernel.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_$c" -o pmc -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-prof > "$OLDPWD/gpurun_out/pmc_$c.log" 2>&1 )
done
python - <<'PY'
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"gpurun_out/pmc_{c}/*counter_collection.csv")
    if not f: print(c,"no csv", glob.glob(f"gpurun_out/pmc_{c}/*")); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name")==c: agg[(r["Kernel_Name"].split("(")[0][:48], r.get("Grid_Size","") )].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:14]:
        print(f"{c:11s} {k[0]:50s} grid={k[1]:>9} n={len(v):3d} avg={sum(v)/len(v):14.1f}")
PY
