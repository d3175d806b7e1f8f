#!/bin/bash
# round 6, evidence visit: the whole GPU suite, then the evidence set r06_final_* (tools/gpu_profiles.sh)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15 | tee gpurun_out/r06_final_gpu_suite.txt
bash tools/gpu_profiles.sh r06_final 2>&1 | tail -80
