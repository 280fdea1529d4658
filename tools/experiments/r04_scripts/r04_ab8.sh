#!/bin/bash
mkdir -p gpurun_out/r04
{
bash tools/ab_bench.sh "C2V79 C2V207 C2V239 C2V79 C2V207" "mid" 524288
} > gpurun_out/r04/ab8.log 2>&1
cat gpurun_out/r04/ab8.log
cp build/ab/libC2V79.so zpaqsharp_amd/libzpaqhip.so
bash tools/prof_stages.sh mid 524288 C2_PROF_MASK0x6000v79 C2_PROF_MASK0x6000v207 > gpurun_out/r04/prof_vm4_mid.log 2>&1; cat gpurun_out/r04/prof_vm4_mid.log
bash tools/prof_l1_miss.sh gpurun_out/r04/stages_l1_miss.txt
for K in T X R; do python3 bench.py --model l1 --kind $K --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('l1', '$K', round(d['value'],1), d['bit_exact'])"; done | tee gpurun_out/r04/l1_txr.txt
