#!/bin/bash
# round 6: the spectral soak again, every failing case printed
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tests/fuzz_gpu.py --spectral --keep-going --seconds 420 --seed 1 > gpurun_out/r06_soak_spectral_seed1.txt 2>&1
grep -c . gpurun_out/r06_soak_spectral_seed1.txt; grep "FAILED\|fuzz" gpurun_out/r06_soak_spectral_seed1.txt | cut -c1-400 | head -40
