for t in "" "--tune 3=1"; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$t', d['value'], d['ms_per_step'], 'seq_fwd', k['seq_fwd']['ms_per_step'], 'seq_bwd', k['seq_bwd']['ms_per_step'])
"; done
