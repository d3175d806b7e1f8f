#!/bin/bash
# usage: quick_bench.sh "cfg2 cfg3 ..." [extra bench args]   -> one "workload clips/s ms" line per run (no CPU baseline, no profiler pass)
W="$1"; shift
for w in $W; do
  echo "$w: $(timeout 300 python bench.py --workload $w --steps 30 --warmup 8 --no-cpu-baseline --no-prof --no-stream-inputs --secondary none "$@" 2>/dev/null | grep '^{' | python -c 'import json,sys;d=json.loads(sys.stdin.read());print(d["value"],d["ms_per_step"])')"
done
