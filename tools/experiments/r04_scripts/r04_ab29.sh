#!/bin/bash
# chain kernels: the whole kernel shifted by 4n bytes (n s_nop at its entry; scratch build, -DC2_SHIFT=n) — what placement alone
# does to compiled code (zh_cm_fast.h rule 2), mid and min at 256 x 1 MiB, same box; then tools/small_blocks.py on the shipped library
mkdir -p gpurun_out/r04
cp zpaqsharp_amd/libzpaqhip.so /tmp/keep.so
for n in 0 1 2 3 4 5 6 7; do
  cp build/ab/libC2S$n.so zpaqsharp_amd/libzpaqhip.so
  for m in mid min; do
    timeout -k 10 300 python3 bench.py --model $m --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shift', $n, '$m', round(d['value'],2), d['bit_exact'])"
  done
done | tee gpurun_out/r04/ab29.txt
cp /tmp/keep.so zpaqsharp_amd/libzpaqhip.so
timeout -k 10 400 python3 tools/small_blocks.py > gpurun_out/r04/small_blocks.txt 2>&1; tail -12 gpurun_out/r04/small_blocks.txt
