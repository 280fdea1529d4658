#!/bin/bash
mkdir -p gpurun_out/r04
bash tools/prof_stages.sh mid 524288 C2_PROF_MASK0x0820v0 C2_PROF_MASK0x0820v63 > gpurun_out/r04/prof_wait_mid.log 2>&1; cat gpurun_out/r04/prof_wait_mid.log
bash tools/prof_stages.sh max+e8e9 524288 C2_PROF_MASK0x0820v0 C2_PROF_MASK0x0820v63 > gpurun_out/r04/prof_wait_max.log 2>&1; cat gpurun_out/r04/prof_wait_max.log
timeout -k 10 300 python tools/small_blocks.py > gpurun_out/r04/small_blocks.txt 2>&1; grep -E " 1024 x| 256 x" gpurun_out/r04/small_blocks.txt
