#!/bin/bash
# L1: second-nibble lane index computed behind the s_addc, three instructions ahead of its v_readlane (ZH_L1_IDX2), same box
mkdir -p gpurun_out/r04
cp build/ab/libZH_L1_IDX21.so zpaqsharp_amd/libzpaqhip.so
bash tools/r04_l1.sh || exit 1
for v in ZH_L1_IDX20 ZH_L1_IDX21 ZH_L1_IDX20 ZH_L1_IDX21; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so
  for K in T X R; do timeout -k 10 120 python3 bench.py --model l1 --kind $K --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$K', round(d['value'],1), d['bit_exact'])"; done
done | tee gpurun_out/r04/ab19.txt
cp build/ab/libZH_L1_IDX21.so zpaqsharp_amd/libzpaqhip.so
