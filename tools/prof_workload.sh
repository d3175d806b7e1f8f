#!/bin/bash
# rocprofv3 kernel stats of one bench workload (outputs -> gpurun_out/<tag>_kernel_stats_<workload>.csv).
# usage: prof_workload.sh <tag> <workload> [steps] ["extra bench args"]
set -u
TAG="$1"; W="$2"; STEPS="${3:-20}"; EXTRA="${4:-}"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_${TAG}_$W" -o $TAG -- python "$OLDPWD/bench.py" --workload $W --steps $STEPS --warmup 5 --no-cpu-baseline --no-prof --no-stream-inputs --secondary none $EXTRA > "$OLDPWD/gpurun_out/${TAG}_rocprof_$W.log" 2>&1 ); echo "rocprof rc=$?"
f=$(find gpurun_out/prof_${TAG}_$W -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats_$W.csv && head -24 "$f"
rm -rf gpurun_out/prof_${TAG}_$W
