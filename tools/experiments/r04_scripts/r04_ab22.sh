#!/bin/bash
# L1: the unchanged loop shifted by 4n bytes against its 256-byte alignment (ZH_L1_PAD = n), text, 256 x 1 MiB, same box, two rounds
mkdir -p gpurun_out/r04
for r in 1 2; do
for n in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
  cp build/ab/libZH_L1_PAD$n.so zpaqsharp_amd/libzpaqhip.so
  timeout -k 10 120 python3 bench.py --model l1 --kind T --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad', $n, round(d['value'],1), d['bit_exact'])"
done; done | tee gpurun_out/r04/ab22.txt
cp build/ab/libZH_L1_PAD0.so zpaqsharp_amd/libzpaqhip.so
