#!/bin/bash
# round 5, call 12: determinism stress at 200 repeats, fuzz with fresh seeds, the GPU suite with the bf16 split on (final build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_determinism.py -m gpu -q --durations=5 > gpurun_out/r05_m_determinism.txt 2>&1; tail -9 gpurun_out/r05_m_determinism.txt
: > gpurun_out/r05_m_fuzz_parity.txt
for s in 11 12; do timeout 300 python tests/fuzz_gpu.py --seconds 120 --seed $s 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_m_fuzz_parity.txt; done
EEG_DCRNN_SPLIT_BF16=1 timeout 200 python tests/fuzz_gpu.py --seconds 90 --seed 13 2>&1 | grep -v amdgpu.ids | sed 's/^/[split-bf16] /' >> gpurun_out/r05_m_fuzz_parity.txt
cat gpurun_out/r05_m_fuzz_parity.txt
EEG_DCRNN_SPLIT_BF16=1 timeout 600 python -m pytest tests -m gpu -q > gpurun_out/r05_m_pytest_gpu_split_mode.txt 2>&1; tail -3 gpurun_out/r05_m_pytest_gpu_split_mode.txt
