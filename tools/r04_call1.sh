#!/bin/bash
# Round-4 GPU visit 1: lane-map / activation lab, parity of the one-value-per-lane remainder kernels, A/B against the round-3
# library (build/ab/base_r03*.so), cycle probes, the driver-protocol line with per-step times.  Outputs -> gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== act_lab"; timeout 120 ./tools/micro/act_lab > $O/r04_act_lab.txt 2>&1; cat $O/r04_act_lab.txt
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/r04_pytest_gpu_1.log 2>&1; tail -12 $O/r04_pytest_gpu_1.log
summ() { python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); k=(d.get('roofline') or {}).get('kernels',{})
    print('$1'.ljust(10), d['value'], d['ms_per_step'], 'p50', d.get('ms_per_step_p50'), 'first5', d.get('first5_ms'), 'last5', d.get('last5_ms'), 'sclk', d.get('shader_clock_mhz_under_load'), ' '.join(f\"{n}={k[n]['ms_per_step']:.4f}\" for n in ('seq_fwd','seq_bwd','gemm_nn_xw','gemm_tn_x') if n in k))"; }
for i in 1 2; do
  echo "== A/B round $i"
  timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-stream-inputs --secondary none 2>/dev/null | tee $O/r04_ab_new_$i.json | summ new
  timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-stream-inputs --secondary none --lib build/ab/base_r03.so 2>/dev/null | tee $O/r04_ab_old_$i.json | summ old
done
echo "== cfg3 / cfg5 (single-wave kernels untouched so far; baseline for later)"
timeout 300 python bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none 2>/dev/null | tee $O/r04_c1_cfg3.json | summ cfg3
echo "== seq_probe new / old"
timeout 200 python tools/seq_probe.py cfg2 > $O/r04_seq_probe_new.txt 2>&1; cat $O/r04_seq_probe_new.txt
timeout 200 python tools/seq_probe.py cfg2 build/ab/base_r03_dev.so > $O/r04_seq_probe_old.txt 2>&1; cat $O/r04_seq_probe_old.txt
echo "== driver protocol (fresh process, --steps 20 --warmup 5), twice"
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/r04_drv_$i.err | tee $O/r04_drv_$i.json | summ drv$i; tail -2 $O/r04_drv_$i.err; done
echo "== long run (300 steps) for the sustained rate"
timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-stream-inputs --secondary none --no-prof 2>/dev/null | tee $O/r04_long.json | summ long
