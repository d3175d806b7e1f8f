#!/bin/bash
# SQ stall breakdown per kernel (counters only + kernel trace): WAVE_CYCLES, WAIT_ANY (parked: waitcnt/barrier),
# WAIT_INST_ANY (issue stall), ACTIVE_INST_ANY, VALU_MFMA_BUSY_CYCLES, LDS bank conflicts.
# usage: pmc_sq.sh [workload] [extra bench args]
W="${1:-cfg2}"; shift
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/sq_$i" -o sq -- \
      python "$OLDPWD/bench.py" --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-graph --no-stream-inputs "$@" > "$OLDPWD/gpurun_out/sq_$i.log" 2>&1 )
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES",
         "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"]
print("kernel".ljust(46) + " ".join(n.replace("SQ_", "")[:14].rjust(15) for n in names))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    if "eeg" not in k: continue
    print(k.ljust(46) + " ".join((f"{sum(d[n])/len(d[n]):15.4g}" if n in d else " " * 15) for n in names))
PY
