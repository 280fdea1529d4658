#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_mid; mkdir -p $OUT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/$T -- python3 bench.py --model mid --blocks 256 --block-bytes 262144 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $OUT/$T.json 2> $OUT/$T.err || { tail -3 $OUT/$T.err; }
  python3 - $OUT/$T <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "zh_decode" in r.get("Kernel_Name", ""):
            print(r["Counter_Name"], r["Counter_Value"])
PY
done
