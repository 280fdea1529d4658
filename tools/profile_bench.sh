#!/bin/bash
# Runs on the GPU box (via gpurun): one bench workload plus the rocprofv3 passes whose summaries are committed under
# profiles/<round>/.  Usage: tools/profile_bench.sh <tag> <model> <blocks> <block-bytes> [more bench args...]
# Outputs in gpurun_out/<tag>/: bench.json (the bench line, taken last), kernel_stats.csv (rocprofv3 --kernel-trace --stats),
# pmc_<model>_<blocks>x<KiB>KiB.json (FETCH_SIZE / WRITE_SIZE per launch + the hash of the kernel sources they belong to).
set -o pipefail
TAG=${1:-run}; MODEL=${2:-l1}; NB=${3:-256}; BS=${4:-4194304}; shift 4
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--model $MODEL --blocks $NB --block-bytes $BS --cache-dir /tmp/zc --no-extras $*"
echo "== kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS --no-cpu-baseline --no-verify > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
cp "$OUT"/trace/*/*_kernel_stats.csv "$OUT/kernel_stats.csv"
head -4 "$OUT/kernel_stats.csv"
# PMC passes, separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains with --pmc)
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-verify > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err" || { tail -5 "$OUT/pmc_$C.err"; exit 1; }
done
python3 tools/pmc_summary.py "$OUT" "$MODEL" "$NB" "$BS"
# the bench line last: with the PMC summary of THESE sources in place (profiles/$ROUND/), bench.py fills roofline.traffic
ROUND=${ROUND:-r03}; mkdir -p profiles/$ROUND; cp "$OUT"/pmc_*KiB.json profiles/$ROUND/
echo "== bench"; python3 bench.py $ARGS > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench.json"
