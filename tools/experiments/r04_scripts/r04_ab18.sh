#!/bin/bash
# L1: bookkeeping in the v_readlane shadows, lag test as one compare, message by one v_lshl_or (ZH_L1_SH2), same box
mkdir -p gpurun_out/r04
cp build/ab/libZH_L1_SH21.so zpaqsharp_amd/libzpaqhip.so
bash tools/r04_l1.sh || exit 1
for v in ZH_L1_SH20 ZH_L1_SH21 ZH_L1_SH20 ZH_L1_SH21; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so
  for K in T X R; do timeout -k 10 120 python3 bench.py --model l1 --kind $K --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$K', round(d['value'],1), d['bit_exact'])"; done
done | tee gpurun_out/r04/ab18.txt
cp build/ab/libZH_L1_SH21.so zpaqsharp_amd/libzpaqhip.so
