#!/bin/bash
# round 5, call 18: GPU suite on the final sources (empty / malformed operand checks), then the evidence set (tag r05_t)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_t_pytest_gpu.txt 2>&1; grep -n "passed\|failed" gpurun_out/r05_t_pytest_gpu.txt | tail -3
timeout 1500 bash tools/gpu_profiles.sh r05_t
