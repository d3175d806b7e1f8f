#!/bin/bash
# HBM traffic per launch of every kernel class from PMC counters, as MI355X_MICROARCH.md prescribes:
# separate passes for FETCH_SIZE and WRITE_SIZE (counters only + kernel trace, no other trace domains);
# read bytes = 2 * FETCH_SIZE * 1024 on gfx950 (wide coalesced reads are tallied at half their size),
# write bytes = WRITE_SIZE * 1024.  Writes gpurun_out/pmc_traffic_<workload>.json (copy into profiles/).
# usage: pmc_traffic.sh [workload] [extra bench args, e.g. --tune 14=64]
W="${1:-cfg2}"; shift
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_${W}_$c" -o pmc -- \
      python "$OLDPWD/bench.py" --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-graph --no-stream-inputs --secondary none "$@" > "$OLDPWD/gpurun_out/pmc_${W}_$c.log" 2>&1 )
  tail -3 "gpurun_out/pmc_${W}_$c.log" | cut -c1-400
done
python - "$W" <<'PY'
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
w = sys.argv[1]
CLASSES = [("seq_fwd", "seq_fwd"), ("seq_bwd", "seq_bwd"), ("gemm_nn", "gemm_nn"), ("gemm_tn", "gemm_tn"),
           ("diffuse_fwd", "diffuse_fwd"), ("diffuse_adj", "diffuse_adj"), ("reduce_unpack", "reduce_unpack"),
           ("corr_gram", "corr_gram"), ("fft200_features", "fft_features"), ("dec_fwd_persist", "dec_fwd_persist"),
           ("dec_bwd_persist", "dec_bwd_persist")]
tot = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = {c: collections.defaultdict(int) for c in ("FETCH_SIZE", "WRITE_SIZE")}
variants = collections.defaultdict(lambda: collections.defaultdict(list))
bysym = collections.defaultdict(lambda: collections.defaultdict(list))


def short(name):
    """rocprofv3's kernel name -> the spelling of bench.py's recorder symbols (bench.short_symbol)"""
    n = name.strip()
    if n.startswith("void "):
        n = n[5:]
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            cut = i
            break
    n = n[:cut]
    if n.startswith("eeg::"):
        n = n[5:]
    return bench.short_symbol(n)



for c in tot:
    files = glob.glob(f"gpurun_out/pmc_{w}_{c}/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no counter csv for", c); sys.exit(1)
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name") != c:
            continue
        name = r["Kernel_Name"]
        bysym[short(name)][c].append(float(r["Counter_Value"]))
        for key, cls in CLASSES:
            if key in name:
                tot[c][cls] += float(r["Counter_Value"]); cnt[c][cls] += 1
                variants[name.split("(")[0][:60]][c].append(float(r["Counter_Value"]))
                break
out = {}
for _, cls in CLASSES:
    if cnt["FETCH_SIZE"][cls] and cnt["WRITE_SIZE"][cls]:
        rd = 2.0 * tot["FETCH_SIZE"][cls] / cnt["FETCH_SIZE"][cls] * 1024
        wr = tot["WRITE_SIZE"][cls] / cnt["WRITE_SIZE"][cls] * 1024
        out[cls] = int(rd + wr)
doc = {"workload": w, "kernel_sources_sha256": bench.kernel_sources_sha256(),
       "note": "HBM bytes per launch (average over all launches of the class) = (2*FETCH_SIZE + WRITE_SIZE)*1024, rocprofv3 --pmc, "
               "separate passes (tools/pmc_traffic.sh); gfx950 correction per MI355X_MICROARCH.md",
       "traffic_bytes_per_launch": out,
       "traffic_bytes_per_launch_by_symbol": {k: int(2.0 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024 + sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024)
                                              for k, d in bysym.items() if d.get("FETCH_SIZE") and d.get("WRITE_SIZE")},
       "per_variant_avg_KB": {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} for k, d in variants.items()}}
json.dump(doc, open(f"gpurun_out/pmc_traffic_{w}.json", "w"), indent=1)
print(json.dumps(doc, indent=1))
PY
