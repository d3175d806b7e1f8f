#!/bin/bash
# Round-5 GPU visit 6: the opt-in three-term bf16 split of the hoisted NN GEMMs: parity (dedicated tests + the WHOLE suite with the mode on), bench second line.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -k "split_bf16" > $O/r05_pytest_gpu_6a.log 2>&1; echo "pytest split rc=$?"; tail -15 $O/r05_pytest_gpu_6a.log
EEG_DCRNN_SPLIT_BF16=1 timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/r05_pytest_gpu_6b_split_mode_on.log 2>&1; echo "pytest (whole suite, split mode on) rc=$?"; tail -12 $O/r05_pytest_gpu_6b_split_mode_on.log
for w in cfg2 cfg3 cfg5; do
  timeout 600 python bench.py --workload $w --split-bf16 --steps 30 --warmup 10 --no-cpu-baseline --no-stream-inputs --secondary none > $O/r05_f_bench_split_bf16_$w.json 2> $O/r05_f_bench_split_bf16_$w.err; echo "bench $w rc=$?"
  python - "$O/r05_f_bench_split_bf16_$w.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); e = d["experimental_split_bf16_step"]
    print("  fp32", d["value"], d["ms_per_step"], "| split", e["clips_per_s"], e["ms_per_step"], "ratio", e["ratio_to_fp32_value"], "| nn gemm ms", e["nn_gemm_ms_per_step_fp32"], "->", e["nn_gemm_ms_per_step_split"], e["kernels"], "loss", e["final_loss_fp32"], e["final_loss"])
except Exception as ex:
    print("  parse failed", ex)
PY
  tail -2 $O/r05_f_bench_split_bf16_$w.err
done
