#!/bin/bash
# Round-4 GPU visit 5: persistent decoder forward with the x-part on the quad pack + register-resident input node mix: parity, phase probe, cfg5 A/B.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "decoder or ssl" > $O/r04_pytest_gpu_5.log 2>&1; tail -4 $O/r04_pytest_gpu_5.log
timeout 300 python tools/dec_probe.py build/ab/decprobe.so 2>&1 | grep -v amdgpu.ids | tee $O/r04_dec_probe_2.txt
bash tools/ab_libs.sh --workload cfg5 --rounds 2 build/ab/cur.so - 2>&1 | tee $O/r04_ab5_cfg5.txt
