#!/bin/bash
# round 6: dev knob 8 = 1 -- the weight-gradient launches of a layer on a side stream beside its input-gradient chain (A/B/A per workload)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for W in cfg2 cfg3 cfg5; do
  for v in 0 1 0 1; do
    echo "$W knob 8=$v: $(timeout 300 python bench.py --workload $W --steps 30 --warmup 10 --no-cpu-baseline --no-stream-inputs --no-prof --secondary none --tune 8=$v 2>gpurun_out/r06_side_$W_$v.err | grep '^{' | python -c 'import json,sys;d=json.loads(sys.stdin.read());print(d["value"],d["ms_per_step"],d["last5_ms"])')"
  done
done
} 2>&1 | tee gpurun_out/r06_ab_knob8_side_stream.txt
tail -3 gpurun_out/r06_side_*.err | tail -20
