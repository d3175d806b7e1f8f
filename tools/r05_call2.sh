#!/bin/bash
# Round-5 GPU visit 2: full GPU suite, driver-protocol bench line (secondary workloads + baselines), raw workload, A/B of the XCD-aware TN placement and the row-streaming adjoint diffusion.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/r05_pytest_gpu_2.log 2>&1; echo "pytest rc=$?"; tail -12 $O/r05_pytest_gpu_2.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r05_b_bench.json 2> $O/r05_b_bench.err; echo "bench rc=$?"; tail -25 $O/r05_b_bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_b_bench.json"))
    print("cfg2", d["value"], d["ms_per_step"], "dominant", d["roofline"]["symbol"], d["roofline"]["frac"], d["roofline"]["ms_per_step"])
    for k, v in d["roofline"]["by_symbol"].items():
        print("   ", k, v["ms_per_step"], v["frac"], v.get("traffic"), v["roles"])
    for k, v in (d.get("secondary_workloads") or {}).items():
        print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "dominant_symbol", "frac", "error", "device_graph_check")})
        for kk, vv in (v.get("top_symbols") or {}).items():
            print("      ", kk, vv)
    print("aten", d.get("aten_gpu_baseline")); print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "sample")})
except Exception as e:
    print("bench parse failed", e)
PY
timeout 300 python bench.py --workload raw --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none > $O/r05_b_bench_raw.json 2> $O/r05_b_bench_raw.err; echo "raw rc=$?"; tail -3 $O/r05_b_bench_raw.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_b_bench_raw.json")); k = d["roofline"]["kernels"]
    print("raw", d["value"], d["ms_per_step"], "fft", k.get("fft_features"), "corr", k.get("corr_gram"), d["config"]["device_graph_check"])
except Exception as e:
    print("raw parse failed", e)
PY
echo "== A/B cfg2: dev defaults | 17=1 plain TN order | 18=1 round-4 adjoint"
bash tools/ab.sh --workload cfg2 "23=0" "17=1" "18=1" "23=0" "17=1" "18=1" 2>&1 | tee $O/r05_b_ab_cfg2.txt
echo "== A/B cfg3"
bash tools/ab.sh --workload cfg3 "23=0" "17=1" "18=1" "23=0" "17=1" "18=1" 2>&1 | tee $O/r05_b_ab_cfg3.txt
