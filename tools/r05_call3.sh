#!/bin/bash
# Round-5 GPU visit 3: featurisation kernel v2 (split logarithm, rotated twiddles), whole-step graph A/B, PMC traffic of the TN GEMM with / without the XCD-aware placement.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -k "fft or raw_signals or full_size_gradients" > $O/r05_pytest_gpu_3.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r05_pytest_gpu_3.log
for r in 1 2; do
for extra in "--workload raw" "" "--graph-update"; do
  timeout 300 python bench.py $extra --steps 30 --warmup 10 --no-cpu-baseline --no-stream-inputs --secondary none 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels']
print('$extra'.ljust(18), d['value'], d['ms_per_step'], 'p50', d['ms_per_step_p50'], d['config']['launch'][:40], '| fft', (k.get('fft_features') or {}).get('ms_per_step'), (k.get('fft_features') or {}).get('frac'), '| corr', (k.get('corr_gram') or {}).get('ms_per_step'))"
done
done 2>&1 | tee $O/r05_c_ab_graph_update.txt
echo "== PMC traffic cfg2, default (k-blocks of a split on one XCD)"
timeout 600 bash tools/pmc_traffic.sh cfg2 > $O/r05_c_pmc_cfg2_default.log 2>&1; cp $O/pmc_traffic_cfg2.json $O/r05_c_pmc_traffic_cfg2_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_c_pmc_traffic_cfg2_default.json")); print({k: v for k, v in d["traffic_bytes_per_launch_by_symbol"].items() if "gemm" in k or "diffuse" in k})
PY
echo "== PMC traffic cfg2, --tune 17=1 (plain workgroup order)"
timeout 600 bash tools/pmc_traffic.sh cfg2 --tune 17=1 > $O/r05_c_pmc_cfg2_plain.log 2>&1; cp $O/pmc_traffic_cfg2.json $O/r05_c_pmc_traffic_cfg2_plain_order.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_c_pmc_traffic_cfg2_plain_order.json")); print({k: v for k, v in d["traffic_bytes_per_launch_by_symbol"].items() if "gemm" in k or "diffuse" in k})
PY
rm -rf $O/pmc_cfg2_FETCH_SIZE $O/pmc_cfg2_WRITE_SIZE
