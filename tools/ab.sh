for t in "" "--tune 2=1" "--tune 2=2"; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$t', d['value'], d['ms_per_step'], 'gemm_nn', k['gemm_nn']['ms_per_step'], 'gemm_tn', k['gemm_tn']['ms_per_step'], 'reduce_unpack', k['reduce_unpack']['ms_per_step'])
"; done
