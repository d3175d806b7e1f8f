#!/bin/bash
# copy the judged files of an evidence visit (tools/gpu_profiles.sh <tag>) from gpurun_out/ into profiles/ (ONE set per round: same names overwrite)
TAG="${1:-r06_final}"; cd "$(dirname "$0")/.."
for f in bench bench_cfg3 bench_cfg4 bench_cfg4_dropout05 bench_cfg5 bench_cfg5_L3 bench_cfg5_curriculum bench_driver_protocol bench_graph_update bench_raw \
         bench_sustained_2000_steps dist1_bench dist1_graph_update_bench; do
  [ -s gpurun_out/${TAG}_$f.json ] && cp gpurun_out/${TAG}_$f.json profiles/
done
for f in gap_stats.txt gpu_suite.txt kernel_stats.csv kernel_stats_cfg3.csv kernel_stats_cfg5.csv kernel_stats_raw.csv scale8_dry_run.txt seq_probe_cfg2.txt sq_stalls_cfg2.txt; do
  [ -s gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/
done
for w in cfg2 cfg3 cfg4 cfg5 raw; do [ -s gpurun_out/pmc_traffic_$w.json ] && cp gpurun_out/pmc_traffic_$w.json profiles/; done
ls profiles | grep -c "${TAG}"
