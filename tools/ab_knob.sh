#!/bin/bash
# A/B of one dev knob inside ONE box visit: ab_knob.sh <workload> <knob> <role in roofline.kernels> v1 v2 ...
W="$1"; K="$2"; ROLE="$3"; shift 3
for v in "$@"; do
  echo "$W knob $K=$v: $(timeout 300 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none --tune $K=$v 2>/dev/null | grep '^{' | python -c 'import json,sys;d=json.loads(sys.stdin.read());k=d["roofline"]["kernels"]["'$ROLE'"];print(d["value"],k["ms_per_step"],k["frac"])')"
done
