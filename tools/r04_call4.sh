#!/bin/bash
# Round-4 GPU visit 4: two-pass GEMM streams in the single-wave recurrent kernels (EEG_SW_TWOPASS bits: 1 fwd gates, 2 fwd candidate,
# 4 bwd GEMM1, 8 bwd GEMM2) and 2 / 4 quads per MFMA-shape group in the streamed-weight GEMMs (EEG_STREAM_QG), cfg3 and cfg5.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== parity of the default build (tp15, qg1) on the M=5 shapes"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "cfg3 or cfg5 or ssl or decoder or cell or stream or sweep" > $O/r04_pytest_gpu_4.log 2>&1; tail -5 $O/r04_pytest_gpu_4.log
echo "== cfg3"
bash tools/ab_libs.sh --workload cfg3 --rounds 2 build/ab/tp0.so build/ab/tp3.so build/ab/tp11.so build/ab/tp15.so 2>&1 | tee $O/r04_ab4_cfg3.txt
echo "== cfg5"
bash tools/ab_libs.sh --workload cfg5 --rounds 2 build/ab/tp0.so build/ab/tp3.so build/ab/tp3qg2.so build/ab/tp3qg4.so 2>&1 | tee $O/r04_ab4_cfg5.txt
