#!/bin/bash
# Round-4 GPU visit 2: whole GPU suite (dropout, L=3), cfg4 with / without dropout, cfg5 at L = 2 / 3.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/r04_pytest_gpu_2.log 2>&1; tail -25 $O/r04_pytest_gpu_2.log
summ() { python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); k=(d.get('roofline') or {}).get('kernels',{})
    print('$1'.ljust(12), d['value'], d['ms_per_step'], 'p50', d.get('ms_per_step_p50'), 'first5', d.get('first5_ms'), ' '.join(f\"{n}={k[n]['ms_per_step']:.4f}\" for n in ('seq_fwd','seq_bwd','cls_head_fwd','cls_head_bwd_dz','cls_head_bwd_w','rng_take','dec_fwd_persist','dec_bwd_persist') if n in k))"; }
for i in 1 2; do
  timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none 2>$O/err | tee $O/r04_cfg4_p0_$i.json | summ cfg4_p0; tail -2 $O/err | grep -v WARNING
  timeout 300 python bench.py --workload cfg4 --dropout 0.5 --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none 2>$O/err | tee $O/r04_cfg4_p05_$i.json | summ cfg4_p0.5; tail -2 $O/err | grep -v WARNING
done
timeout 300 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none 2>$O/err | tee $O/r04_cfg5_L2.json | summ cfg5_L2; tail -2 $O/err | grep -v WARNING
timeout 300 python bench.py --workload cfg5 --dropout 0.5 --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none 2>$O/err | tee $O/r04_cfg5_L2_p05.json | summ cfg5_L2_p.5; tail -2 $O/err | grep -v WARNING
timeout 300 python bench.py --workload cfg5 --layers 3 --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none 2>$O/err | tee $O/r04_cfg5_L3.json | summ cfg5_L3; tail -2 $O/err | grep -v WARNING
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none 2>$O/err | tee $O/r04_cfg2_c2.json | summ cfg2; tail -2 $O/err | grep -v WARNING
