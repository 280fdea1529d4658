#!/bin/bash
# L1: t in a VGPR, chunk test folded into the lag compare, EOS renormalisation test by the carry into the top byte (E1), same box
mkdir -p gpurun_out/r04
cp build/ab/libE1.so zpaqsharp_amd/libzpaqhip.so
bash tools/r04_l1.sh || exit 1
for v in E0 E1 E0 E1; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so
  for K in T X R; do timeout -k 10 120 python3 bench.py --model l1 --kind $K --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$K', round(d['value'],1), d['bit_exact'])"; done
done | tee gpurun_out/r04/ab20.txt
cp build/ab/libE1.so zpaqsharp_amd/libzpaqhip.so
