#!/bin/bash
# round 6, visit 1: spectral-form parity at full size + A/B of the cfg2 step (spectral vs general path) in one call
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "spectral_form or test_full_size_gradients_vs_oracle" 2>&1 | tail -15
for mode in 1 0 1 0; do
  echo "== cfg2 EEG_DCRNN_SPECTRAL=$mode"
  EEG_DCRNN_SPECTRAL=$mode timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-stream-inputs --secondary none > gpurun_out/r06_1_bench_spec$mode.json 2> gpurun_out/r06_1_bench_spec$mode.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_1_bench_spec$mode.json").read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms; final loss", d["config"]["final_loss"])
r = d.get("roofline") or {}
for k, v in (r.get("kernels") or {}).items():
    print(f"   {k:16s} {v['ms_per_step']:.4f} ms  x{v['launches_per_step']:.0f}  frac {v.get('frac')}  {v.get('symbol','')[:60]}")
PY
  tail -2 gpurun_out/r06_1_bench_spec$mode.err
done
