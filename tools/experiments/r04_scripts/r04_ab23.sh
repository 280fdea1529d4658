#!/bin/bash
# L1: the byte laid out so that every not-taken branch and the instruction behind it share a 32-byte fetch window
# (64-byte steps, v_nop in the v_readlane shadows as filler; ZH_L1_PAD=5) against the old layout at its best shift, same box
mkdir -p gpurun_out/r04
cp build/ab/libZH_L1_PAD5n.so zpaqsharp_amd/libzpaqhip.so
bash tools/r04_l1.sh || exit 1
for v in OLDPAD1 ZH_L1_PAD5n OLDPAD1 ZH_L1_PAD5n; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so
  for K in T X R; do timeout -k 10 120 python3 bench.py --model l1 --kind $K --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$K', round(d['value'],1), d['bit_exact'])"; done
done | tee gpurun_out/r04/ab23.txt
for v in ZH_L1_PAD3n ZH_L1_PAD4n ZH_L1_PAD6n ZH_L1_PAD7n; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so
  timeout -k 10 120 python3 bench.py --model l1 --kind T --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'T', round(d['value'],1), d['bit_exact'])"
done | tee -a gpurun_out/r04/ab23.txt
cp build/ab/libZH_L1_PAD5n.so zpaqsharp_amd/libzpaqhip.so
