#!/bin/bash
# lazy2: lz_window's assembly loop shifted by 4n bytes inside its 32-byte fetch windows (ZH_LZ2_PAD = n), 256 distinct x 4 MiB, same box
mkdir -p gpurun_out/r04
cp zpaqsharp_amd/libzpaqhip.so /tmp/keep.so
for n in 0 1 2 3 4 5 6 7; do
  cp build/ab/libZH_LZ2_PAD$n.so zpaqsharp_amd/libzpaqhip.so
  timeout -k 10 200 python tools/method_rate.py --methods "x2,1,4,0,3,22" 2>/dev/null | sed "s/^/pad $n: /"
done | tee gpurun_out/r04/ab27.txt
cp /tmp/keep.so zpaqsharp_amd/libzpaqhip.so
