#!/bin/bash
# Round-5 GPU visit 8: randomized parity of the final build (fp32 mode, 3 seeds; opt-in bf16 split mode, 1 seed).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_h_fuzz_parity.txt; : > $O
for seed in 11 12 13; do timeout 400 python tests/fuzz_gpu.py --seconds 150 --seed $seed 2>&1 | grep -v amdgpu.ids | tee -a $O; done
echo "-- EEG_DCRNN_SPLIT_BF16=1 (the hoisted NN GEMMs of 64-unit encoder layers as a three-term bf16 split)" | tee -a $O
EEG_DCRNN_SPLIT_BF16=1 timeout 400 python tests/fuzz_gpu.py --seconds 150 --seed 14 2>&1 | grep -v amdgpu.ids | tee -a $O
