#!/bin/bash
# Round-4 GPU visit 3: suite after the ATen-free step + L1 remainder in the single-wave kernels; cfg2/3/5 lines; kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/r04_pytest_gpu_3.log 2>&1; tail -8 $O/r04_pytest_gpu_3.log
summ() { python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); k=(d.get('roofline') or {}).get('kernels',{})
    print('$1'.ljust(12), d['value'], d['ms_per_step'], 'p50', d.get('ms_per_step_p50'), 'first5', d.get('first5_ms'), ' '.join(f\"{n}={k[n]['ms_per_step']:.4f}\" for n in ('seq_fwd','seq_bwd','dec_fwd_persist','dec_bwd_persist') if n in k), 'ktot', (d.get('roofline') or {}).get('kernel_ms_per_step_total'))"; }
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none 2>$O/err | tee $O/r04_c3_cfg2_$i.json | summ cfg2; done
timeout 300 python bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none 2>$O/err | tee $O/r04_c3_cfg3.json | summ cfg3
timeout 300 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none 2>$O/err | tee $O/r04_c3_cfg5.json | summ cfg5
echo "== rocprofv3 kernel stats cfg2"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof_c3" -o c3 -- python "$OLDPWD/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-prof --no-stream-inputs --secondary none > "$OLDPWD/$O/r04_c3_rocprof.log" 2>&1 ); echo "rocprof rc=$?"
f=$(find $O/prof_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r04_c3_kernel_stats.csv && head -30 "$f" | cut -c1-150
find $O/prof_c3 -name "*kernel_trace.csv" -delete
