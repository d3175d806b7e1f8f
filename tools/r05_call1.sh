#!/bin/bash
# Round-5 GPU visit 1: ABI v4 (device teacher flags, whole-step graph, SSL eval), W=200 mixed-radix featurisation, bench line with secondary workloads.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/r05_smoke.log
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/r05_pytest_gpu_1.log 2>&1; echo "pytest rc=$?"; tail -25 $O/r05_pytest_gpu_1.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r05_a_bench.json 2> $O/r05_a_bench.err; echo "bench rc=$?"; tail -12 $O/r05_a_bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_a_bench.json"))
    print("cfg2", d["value"], d["ms_per_step"], "dominant", d["roofline"]["symbol"], d["roofline"]["frac"], d["roofline"]["ms_per_step"])
    for k, v in d["roofline"]["by_symbol"].items():
        print("   ", k, v["ms_per_step"], v["frac"], v.get("traffic"))
    for k, v in (d.get("secondary_workloads") or {}).items():
        print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "dominant_symbol", "frac", "error")})
        for kk, vv in (v.get("top_symbols") or {}).items():
            print("      ", kk, vv)
    print("aten", d.get("aten_gpu_baseline")); print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "sample")})
except Exception as e:
    print("bench parse failed", e)
PY
for extra in "--workload raw" "--workload cfg5 --curriculum" "--graph-update" "--force-dist --graph-update" "--workload cfg4 --dropout 0.5 --graph-update"; do
  tag=$(echo "$extra" | tr -d ' -' | tr '.' 'p')
  timeout 300 python bench.py $extra --steps 20 --warmup 5 --no-cpu-baseline --no-stream-inputs --secondary none > $O/r05_a_bench_$tag.json 2> $O/r05_a_bench_$tag.err; echo "bench $extra rc=$?"
  python - "$O/r05_a_bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d["roofline"]["kernels"]
    print("  ", d["value"], d["ms_per_step"], d["config"]["launch"], "| fft", k.get("fft_features"), "| teacher", k.get("dec_teacher_flags") or k.get("teacher_flags"))
except Exception as e:
    print("  parse failed", e)
PY
  tail -2 $O/r05_a_bench_$tag.err
done
