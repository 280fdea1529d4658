#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_wave_cm or more_single_cm or golden or sweep or l1 or cm" > gpurun_out/r04/l1miss_tests.log 2>&1; tail -5 gpurun_out/r04/l1miss_tests.log
timeout -k 10 300 python tests/fuzz_l1.py 16 11 0 > gpurun_out/r04/fuzz_a.log 2>&1; tail -2 gpurun_out/r04/fuzz_a.log
timeout -k 10 300 python tests/fuzz_l1.py 12 12 6 > gpurun_out/r04/fuzz_b.log 2>&1; tail -2 gpurun_out/r04/fuzz_b.log
for K in T X R; do python3 bench.py --model l1 --kind $K --blocks 256 --block-bytes 1048576 --no-extras --no-cpu-baseline --cache-dir /tmp/zc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('l1', '$K', round(d['value'],1), d['bit_exact'])"; done | tee gpurun_out/r04/l1_txr2.txt
bash tools/prof_l1_miss.sh gpurun_out/r04/stages_l1_miss2.txt
