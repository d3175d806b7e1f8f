#!/bin/bash
# quick GPU visit: selected tests + bench variants.  usage: gpu_quick.sh "<pytest -k expr>" [bench args...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
K="$1"; shift
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "$K" 2>&1 | tail -15
for v in "$@"; do
  echo "== bench $v"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $v 2> gpurun_out/bench_q.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline') or {}
    print(d['value'], d['unit'], d['ms_per_step'], 'ms', d['config'].get('launch'), 'kernel_total', r.get('kernel_ms_per_step_total'))
"; tail -3 gpurun_out/bench_q.err
done
