#!/bin/bash
# The exact commands of a SCALE run on one 8-GPU MI355X node (one process per GPU over RCCL/xGMI, weak scaling: 256 clips per
# GPU for cfg2/cfg3/cfg4, 512 for cfg5) -- what the driver launches for N = 1, 2, 4, 8 -- for every BASELINE workload that is
# data-parallel (cfg4: 4-class, batch 2048 over 8; cfg5: SSL, batch 4096 over 8; cfg2: the metric's config).
#   tools/scale8.sh                  # N = 1 2 4 8, workloads cfg2 cfg4 cfg5, eager exchange + optimiser tail behind the step graph
#   NPROCS="1" tools/scale8.sh       # dry run on a 1-GPU box (what the builder could execute: world-size-1 RCCL group via --force-dist)
#   EXTRA="--graph-update" ...       # the all-reduce + clip/Adam captured into the step's HIP graph (ONE launch per rank and step)
# Each run prints the bench line of rank 0 (value = aggregate clips/s over all N GPUs, MAX over ranks of the timed region).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0          # dmabuf IPC (RCCL across processes needs it on this image)
NPROCS="${NPROCS:-1 2 4 8}"; WORKLOADS="${WORKLOADS:-cfg2 cfg4 cfg5}"; EXTRA="${EXTRA:-}"
STEPS="${STEPS:-30}"; WARMUP="${WARMUP:-10}"; PORT=29611
for w in $WORKLOADS; do
  for n in $NPROCS; do
    PORT=$((PORT + 1))
    out=gpurun_out/scale_${w}_n${n}.json
    if [ "$n" == "1" ]; then
      # (N = 1 is a plain process for the driver; --force-dist adds a world-size-1 RCCL group so that the exchange is issued)
      python bench.py --gpus 1 --workload $w --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-stream-inputs --secondary none --force-dist $EXTRA > $out 2> ${out%.json}.err
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $n --workload $w --steps $STEPS --warmup $WARMUP --no-stream-inputs $EXTRA > $out 2> ${out%.json}.err
    fi
    python - "$out" "$w" "$n" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]} N={sys.argv[3]}: {d['value']:.0f} clips/s aggregate, {d['ms_per_step']} ms/step (max over ranks), per rank {d['distributed']['per_rank_ms_per_step']}, "
          f"backend {d['distributed']['backend']}, exchange+update {d['distributed']['reduce_and_update_ms_per_step']} ms, launch: {d['config']['launch']}")
except Exception as e:
    print(f"{sys.argv[2]} N={sys.argv[3]}: no bench line ({e}); see {sys.argv[1][:-5]}.err")
PY
  done
done
