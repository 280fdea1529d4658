#!/bin/bash
# Runs on the GPU box (via gpurun): the headline bench plus the rocprofv3 passes whose
# summaries are committed under profiles/.  Usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== bench"; python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench.json"
echo "== kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py "$@" --no-cpu-baseline --no-verify > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
cp "$OUT"/trace/*/*_kernel_stats.csv "$OUT/kernel_stats.csv"
head -4 "$OUT/kernel_stats.csv"
# PMC passes, separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains with --pmc)
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-verify > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err" || { tail -5 "$OUT/pmc_$C.err"; exit 1; }
  python3 - "$OUT/pmc_$C" $C <<'PY'
import csv, glob, sys
d, c = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/*/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == c and "zh_decode" in r.get("Kernel_Name", "")]
    for r in rows:
        print(c, r["Kernel_Name"], r["Counter_Value"])
    with open(d + "_summary.csv", "w") as o:
        o.write("kernel,counter,value\n")
        for r in rows:
            o.write(f'{r["Kernel_Name"]},{c},{r["Counter_Value"]}\n')
PY
done
