#!/bin/bash
# Round-5 GPU visit 4: full suite on the bias-merge build; A/B previous commit vs current (cfg2, cfg3); featurisation kernel at 3 waves per SIMD; correlation-Gram split counts.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/r05_pytest_gpu_4.log 2>&1; echo "pytest rc=$?"; tail -6 $O/r05_pytest_gpu_4.log
echo "== A/B cfg2: previous commit (separate reduce_bias launch) vs current"
bash tools/ab_libs.sh --workload cfg2 --rounds 3 build/ab/prev.so - 2>&1 | tee $O/r05_d_ab_bias_merge_cfg2.txt
echo "== A/B raw: featurisation kernel compiled for 3 waves per SIMD (168 registers, 9 spilled) vs 2 (176)"
for r in 1 2; do for lib in "-" "build/ab/fft3.so"; do
  a=""; [ "$lib" != "-" ] && a="--lib $lib"
  timeout 300 python bench.py --workload raw $a --steps 30 --warmup 10 --no-cpu-baseline --no-stream-inputs --secondary none 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels']
print('$lib'.ljust(22), d['value'], d['ms_per_step'], 'fft', k['fft_features']['ms_per_step'], k['fft_features']['frac'])"
done; done 2>&1 | tee $O/r05_d_ab_fft_occupancy.txt
echo "== correlation Gram: workgroup target (dev knob 7; default 1024 -> 4 splits of a clip at B = 256, 2 at B = 512)"
bash tools/ab.sh --workload cfg3 "23=0" "7=768" "7=1280" "7=1536" "23=0" "7=1280" 2>&1 | tee $O/r05_d_ab_corr_gram_cfg3.txt
bash tools/ab.sh --workload cfg5 "23=0" "7=1536" "7=2048" "23=0" "7=1536" 2>&1 | tee $O/r05_d_ab_corr_gram_cfg5.txt
