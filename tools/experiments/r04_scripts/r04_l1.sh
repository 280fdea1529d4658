#!/bin/bash
# quick L1 check on the GPU box: parity tests of the single-CM kernel, fuzz, then MB/s on the three plaintexts (256 x 1 MiB)
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_wave_cm or more_single_cm or golden or sweep or l1 or cm" > gpurun_out/r04/l1miss_tests.log 2>&1; tail -3 gpurun_out/r04/l1miss_tests.log
grep -q passed gpurun_out/r04/l1miss_tests.log && ! grep -q failed gpurun_out/r04/l1miss_tests.log || exit 1
timeout -k 10 200 python tests/fuzz_l1.py 16 31 0 > gpurun_out/r04/fuzz_a.log 2>&1; tail -1 gpurun_out/r04/fuzz_a.log
timeout -k 10 200 python tests/fuzz_l1.py 16 32 6 > gpurun_out/r04/fuzz_b.log 2>&1; tail -1 gpurun_out/r04/fuzz_b.log
for K in T X R; do timeout -k 10 120 python3 bench.py --model l1 --kind $K --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('l1', '$K', round(d['value'],1), d['bit_exact'])"; done | tee gpurun_out/r04/l1_txr2.txt
