#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, bench, rocprofv3 kernel stats.  Outputs -> gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
{ nproc; lscpu | grep -E "Model name|Socket|Core|Thread"; rocm-smi --showproductname 2>/dev/null | head -8; } > gpurun_out/env.log 2>&1
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.log; tail -5 gpurun_out/bench.err
if [ "${1:-}" != "noprof" ]; then
  echo "== rocprofv3"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o r02 -- python "$OLDPWD/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-prof --no-stream-inputs --secondary none --secondary none > "$OLDPWD/gpurun_out/rocprof.log" 2>&1 ); echo "rocprof rc=$?"
  find gpurun_out/prof -name "*kernel_stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
  # keep only the small summaries (the traces are large)
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
