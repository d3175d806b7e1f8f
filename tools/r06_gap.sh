#!/bin/bash
# round 6: kernel trace of the cfg2 step -> idle time / launches per step (tools/gap_stats.py)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/prof_gap" -o gap -- python "$OLDPWD/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-prof --no-stream-inputs --secondary none > "$OLDPWD/gpurun_out/gap_rocprof.log" 2>&1 )
t=$(find gpurun_out/prof_gap -name "*kernel_trace.csv" | head -1); python tools/gap_stats.py "$t" | tee gpurun_out/r06_final_gap_stats.txt
rm -rf gpurun_out/prof_gap
