#!/bin/bash
mkdir -p gpurun_out/r04
tools/ubench/sgpr_bench > gpurun_out/r04/sgpr_bench.txt 2>&1; cat gpurun_out/r04/sgpr_bench.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_wave_cm or more_single_cm or golden or sweep" > gpurun_out/r04/l1x2_tests.log 2>&1; tail -5 gpurun_out/r04/l1x2_tests.log
timeout -k 10 300 python tests/fuzz_l1.py 12 7 6 > gpurun_out/r04/fuzz_x2.log 2>&1; tail -2 gpurun_out/r04/fuzz_x2.log
timeout -k 10 600 python tools/small_blocks.py > gpurun_out/r04/small_blocks.txt 2>&1; cat gpurun_out/r04/small_blocks.txt
