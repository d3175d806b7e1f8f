#!/bin/bash
# One GPU-box visit that regenerates the evidence under profiles/ (outputs -> gpurun_out/, copy what is to be judged).
# usage: gpu_profiles.sh <tag>      e.g. r03_c
set -u
TAG="${1:-r03}"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
# the PMC pass first: the bench lines below then carry `roofline.traffic` measured on exactly these kernel sources (stamp check in bench.py)
echo "== PMC traffic"; timeout 900 bash tools/pmc_traffic.sh cfg2 > gpurun_out/${TAG}_pmc_traffic.log 2>&1; tail -3 gpurun_out/${TAG}_pmc_traffic.log; cp gpurun_out/pmc_traffic_cfg2.json profiles/pmc_traffic_cfg2.json
echo "== bench cfg2 (driver defaults)"; timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -2 gpurun_out/${TAG}_bench.err
echo "== bench --force-dist"; timeout 300 python bench.py --force-dist --no-cpu-baseline --secondary none --steps 30 --warmup 10 2> gpurun_out/${TAG}_dist1.err | grep '^{' > gpurun_out/${TAG}_dist1_bench.json; tail -2 gpurun_out/${TAG}_dist1.err
for w in cfg3 cfg4 cfg5 raw; do
  echo "== PMC traffic $w"; timeout 900 bash tools/pmc_traffic.sh $w > gpurun_out/${TAG}_pmc_traffic_$w.log 2>&1; cp gpurun_out/pmc_traffic_$w.json profiles/pmc_traffic_$w.json
done
# the driver's line once more, now with the traffic files of every workload in place (secondary_workloads carry `traffic`)
echo "== bench cfg2 (driver protocol: --steps 20 --warmup 5)"; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_protocol.json 2> gpurun_out/${TAG}_bench_driver_protocol.err; tail -2 gpurun_out/${TAG}_bench_driver_protocol.err
for w in cfg3 cfg4 cfg5 raw; do
  echo "== bench $w"; timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err; tail -1 gpurun_out/${TAG}_bench_$w.err
done
echo "== bench cfg5 --curriculum / cfg2 --graph-update / --force-dist --graph-update"
timeout 600 python bench.py --workload cfg5 --curriculum --steps 20 --warmup 5 --no-cpu-baseline --secondary none > gpurun_out/${TAG}_bench_cfg5_curriculum.json 2> gpurun_out/${TAG}_bench_cfg5_curriculum.err
timeout 600 python bench.py --graph-update --steps 20 --warmup 5 --no-cpu-baseline --secondary none > gpurun_out/${TAG}_bench_graph_update.json 2> gpurun_out/${TAG}_bench_graph_update.err
timeout 600 python bench.py --force-dist --graph-update --steps 20 --warmup 5 --no-cpu-baseline --secondary none 2> gpurun_out/${TAG}_dist1_graph_update.err | grep '^{' > gpurun_out/${TAG}_dist1_graph_update_bench.json
echo "== scale8.sh dry run (one process, world-size-1 RCCL group)"; NPROCS=1 bash tools/scale8.sh > gpurun_out/${TAG}_scale8_dry_run.txt 2>&1; cat gpurun_out/${TAG}_scale8_dry_run.txt
echo "== bench cfg4 --dropout 0.5 (README.md:83) / cfg5 --layers 3 (README.md:91)"
timeout 600 python bench.py --workload cfg4 --dropout 0.5 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg4_dropout05.json 2> gpurun_out/${TAG}_bench_cfg4_dropout05.err
timeout 600 python bench.py --workload cfg5 --layers 3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg5_L3.json 2> gpurun_out/${TAG}_bench_cfg5_L3.err
echo "== sustained rate (2000 steps)"; timeout 600 python bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-prof --no-stream-inputs --secondary none > gpurun_out/${TAG}_bench_sustained_2000_steps.json 2>/dev/null
echo "== cycle probe"; timeout 300 python tools/seq_probe.py cfg2 > gpurun_out/${TAG}_seq_probe_cfg2.txt 2>&1
for w in cfg3 cfg5 raw; do bash tools/prof_workload.sh $TAG $w 20 > gpurun_out/${TAG}_prof_$w.log 2>&1; done
echo "== rocprofv3 kernel stats"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_$TAG" -o $TAG -- python "$OLDPWD/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-prof --no-stream-inputs --secondary none > "$OLDPWD/gpurun_out/${TAG}_rocprof.log" 2>&1 ); echo "rocprof rc=$?"
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats.csv && head -14 "$f"
t=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/gap_stats.py "$t" > gpurun_out/${TAG}_gap_stats.txt 2>&1 && head -3 gpurun_out/${TAG}_gap_stats.txt
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
echo "== SQ stalls"; timeout 900 bash tools/pmc_sq.sh cfg2 > gpurun_out/${TAG}_sq_stalls_cfg2.txt 2>&1; head -12 gpurun_out/${TAG}_sq_stalls_cfg2.txt
for d in gpurun_out/pmc_cfg?_FETCH_SIZE gpurun_out/pmc_cfg?_WRITE_SIZE gpurun_out/pmc_raw_FETCH_SIZE gpurun_out/pmc_raw_WRITE_SIZE gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3 gpurun_out/prof_$TAG; do rm -rf "$d"; done   # (the directories; the *.log files stay)
