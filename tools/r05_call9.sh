#!/bin/bash
# round 5, call 9: GPU suite on the race-fixed build, then the evidence set (tag r05_j)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_j_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r05_j_pytest_gpu.txt
timeout 1500 bash tools/gpu_profiles.sh r05_j
