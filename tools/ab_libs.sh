#!/bin/bash
# A/B of library builds inside ONE GPU visit: ab_libs.sh [--workload W] [--rounds R] lib1.so lib2.so ...  ("-" = the product library)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
WL=cfg2; R=2
while [[ "${1:-}" == --* ]]; do case "$1" in --workload) WL=$2; shift 2;; --rounds) R=$2; shift 2;; esac; done
for r in $(seq 1 $R); do
  for t in "$@"; do
    args=""; [ "$t" != "-" ] && args="--lib $t"
    python bench.py --workload $WL --steps 30 --warmup 10 --no-cpu-baseline --no-stream-inputs --secondary none $args 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); k=d['roofline']['kernels']
    print('$t'.ljust(28), d['value'], d['ms_per_step'], 'p50', d.get('ms_per_step_p50'), ' '.join(f\"{n}={k[n]['ms_per_step']:.4f}\" for n in ('gemm_dx_f','spec_mix_dx','seq_fwd','seq_bwd','gemm_nn_xw','gemm_nn_dx','gemm_tn_f','gemm_tn_x','gemm_tn_h','gemm_tn_hg','gemm_tn_hc','dec_fwd_persist','dec_bwd_persist','diffuse_fwd','diffuse_adj','corr_gram') if n in k), ' '.join(f\"{a.split('<')[-1]}={b['ms_per_step']:.4f}\" for a,b in [x for r in ('gemm_tn_f','gemm_nn_xw') for x in (k.get(r,{}).get('by_symbol') or {}).items()]))"
  done
done
