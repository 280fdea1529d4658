#!/bin/bash
# L1: two of call 20's three changes alone — E2 the EOS flag's renormalisation test by the carry into the top byte, E3 t in a VGPR — same box
mkdir -p gpurun_out/r04
for v in E2 E3; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_wave_cm or more_single_cm or golden or sweep or l1 or cm" > gpurun_out/r04/ab21_tests_$v.log 2>&1; tail -1 gpurun_out/r04/ab21_tests_$v.log
  grep -q passed gpurun_out/r04/ab21_tests_$v.log && ! grep -q failed gpurun_out/r04/ab21_tests_$v.log || exit 1
done
for v in E0 E2 E3 E0 E2 E3; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so
  for K in T X R; do timeout -k 10 120 python3 bench.py --model l1 --kind $K --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$K', round(d['value'],1), d['bit_exact'])"; done
done | tee gpurun_out/r04/ab21.txt
cp build/ab/libE0.so zpaqsharp_amd/libzpaqhip.so
