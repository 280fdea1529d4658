#!/bin/bash
# On the GPU box: instruction-mix PMC passes (VALU / SALU / LDS / VMEM / SMEM / wave cycles) over the decode kernel of one
# model, three separate rocprofv3 --pmc runs (no trace domains with --pmc).
# Usage: tools/pmc_mix.sh <model> <block-bytes> ; writes gpurun_out/instmix_<model>/instmix_<model>_256x<KiB>KiB.txt
MODEL=${1:-mid}; BS=${2:-262144}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$(echo $MODEL | tr '+' '_')
OUT=gpurun_out/instmix_$TAG; mkdir -p $OUT
TXT=$OUT/instmix_${TAG}_256x$((BS >> 10))KiB.txt
python3 -c "import bench; print('src_hash', bench.source_hash('$MODEL'))" > $TXT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/$T -- python3 bench.py --model $MODEL --blocks 256 --block-bytes $BS --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-verify --cache-dir /tmp/zc > $OUT/$T.json 2> $OUT/$T.err || { tail -3 $OUT/$T.err; }
  python3 - $OUT/$T <<'PY' >> $TXT
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "zh_decode" in r.get("Kernel_Name", ""):
            print(r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"])
PY
done
echo "decoded bits per launch: $((256 * BS * 8))" >> $TXT
cat $TXT
