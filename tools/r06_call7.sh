#!/bin/bash
# round 6: more soak on the final sources (other seeds)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for sd in 2 3; do
  timeout 600 python tests/fuzz_gpu.py --spectral --keep-going --seconds 400 --seed $sd > gpurun_out/r06_soak_spectral_seed$sd.txt 2>&1
  grep "FAILED\|fuzz\|kink" gpurun_out/r06_soak_spectral_seed$sd.txt | cut -c1-400 | head -20
done
timeout 500 python tests/fuzz_gpu.py --seconds 360 --seed 8 > gpurun_out/r06_soak_general_seed8.txt 2>&1; tail -3 gpurun_out/r06_soak_general_seed8.txt | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_determinism.py -q 2>&1 | tail -2
