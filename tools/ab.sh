#!/bin/bash
# A/B timing of tuning-knob sets (dev build): ab.sh [--workload W] "k=v,k=v" "k=v" ...   ("-" = product defaults)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
WL=cfg2; if [ "$1" == "--workload" ]; then WL=$2; shift 2; fi
for t in "$@"; do
  args=""; if [ "$t" != "-" ]; then for kv in ${t//,/ }; do args="$args --tune $kv"; done; fi
  python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']
print('$t'.ljust(14), d['value'], d['ms_per_step'], ' '.join(f\"{n}={k[n]['ms_per_step']:.4f}\" for n in ('gemm_dx_f','spec_mix_dx','gemm_tn_f','gemm_nn_xw','gemm_nn_dx','seq_fwd','seq_bwd','reduce_unpack','diffuse_fwd','diffuse_adj','corr_gram','dec_fwd_persist','dec_bwd_persist','dec_seq_bwd','dec_gemm_nn_dx','dec_diffuse_adj','dec_reduce_bias') if n in k))"
done
