#!/bin/bash
# round 6, soak + A/B visit: long seeded draws through the spectral form and the general path, then dev knob 8 (forward streaming
# diffusion: passes-fastest workgroup order / clips per launch) at the cfg5 and cfg3 shapes
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tests/fuzz_gpu.py --spectral --seconds 540 --seed 1 2>&1 | tail -6 | tee gpurun_out/r06_soak_spectral.txt
timeout 500 python tests/fuzz_gpu.py --seconds 300 --seed 7 2>&1 | tail -6 | tee gpurun_out/r06_soak_general.txt
{
bash tools/ab_knob.sh cfg5 8 diffuse_fwd 0 1 512 513 256 257 0
bash tools/ab_knob.sh cfg3 8 diffuse_fwd 0 1 256 257 0
} 2>&1 | tee gpurun_out/r06_ab_knob8_diffuse.txt
