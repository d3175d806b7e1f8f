#!/bin/bash
# round 6, visit 2: the whole GPU suite on the spectral build + the driver-protocol bench line
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -15
echo "== bench (driver protocol)"
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_2_bench.json 2> gpurun_out/r06_2_bench.err
tail -5 gpurun_out/r06_2_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r06_2_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms; final loss", d["config"]["final_loss"])
print({k: (v["value"], v["ms_per_step"]) for k, v in (d.get("secondary_workloads") or {}).items()})
print("cpu", d.get("cpu_baseline"))
PY
