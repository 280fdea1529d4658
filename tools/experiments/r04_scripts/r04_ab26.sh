#!/bin/bash
# L1: the 672-byte loop shifted by 4n bytes (ZH_L1_PAD = n), text and random plaintext, same box
mkdir -p gpurun_out/r04
for n in 0 1 2 3 4 5 6 7; do
  cp build/ab/libZH_L1_PAD${n}f.so zpaqsharp_amd/libzpaqhip.so
  for K in T R; do timeout -k 10 120 python3 bench.py --model l1 --kind $K --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad', $n, '$K', round(d['value'],1), d['bit_exact'])"; done
done | tee gpurun_out/r04/ab26.txt
cp build/ab/libZH_L1_PAD5f.so zpaqsharp_amd/libzpaqhip.so
