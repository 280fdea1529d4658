#!/bin/bash
# L1: call 20's three changes (t in a VGPR, chunk test folded into the lag compare, EOS renormalisation test by the carry into
# the top byte) on the 64-byte-step layout — the byte is 672 bytes, no filler in the lookup shadow — against the shipped loop, same box
mkdir -p gpurun_out/r04
cp build/ab/libZH_L1_PAD5f.so zpaqsharp_amd/libzpaqhip.so
bash tools/r04_l1.sh || exit 1
for v in F0 ZH_L1_PAD5f F0 ZH_L1_PAD5f; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so
  for K in T X R; do timeout -k 10 120 python3 bench.py --model l1 --kind $K --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$K', round(d['value'],1), d['bit_exact'])"; done
done | tee gpurun_out/r04/ab25.txt
for v in ZH_L1_PAD2f ZH_L1_PAD6f; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so
  timeout -k 10 120 python3 bench.py --model l1 --kind T --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'T', round(d['value'],1), d['bit_exact'])"
done | tee -a gpurun_out/r04/ab25.txt
cp build/ab/libZH_L1_PAD5f.so zpaqsharp_amd/libzpaqhip.so
