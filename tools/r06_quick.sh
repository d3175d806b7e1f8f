#!/bin/bash
# round 6 quick visit: usage r06_quick.sh "<pytest -k expr or ''>" [bench arg sets...]   (each bench: 30 steps, per-kernel table)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
K="$1"; shift
if [ -n "$K" ]; then timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 -k "$K" 2>&1 | tail -8; fi
i=0
for v in "$@"; do
  i=$((i+1))
  echo "== bench $v"
  eval "timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-stream-inputs --secondary none $v" > gpurun_out/q_bench_$i.json 2> gpurun_out/q_bench_$i.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/q_bench_$i.json").read().strip().splitlines()[-1])
    print(d["value"], d["unit"], d["ms_per_step"], "ms; final loss", d["config"]["final_loss"])
    r = d.get("roofline") or {}
    for k, v in (r.get("kernels") or {}).items():
        if v["ms_per_step"] >= 0.02:
            print(f"   {k:16s} {v['ms_per_step']:.4f} ms  x{v['launches_per_step']:.0f}  frac {v.get('frac')}  {v.get('symbol','')[:70]}")
    print("   kernels total", r.get("kernel_ms_per_step_total"))
except Exception as e:
    print("bench failed:", e); print(open("gpurun_out/q_bench_$i.err").read()[-1500:])
PY
done
