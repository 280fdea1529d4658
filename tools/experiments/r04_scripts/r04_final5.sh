#!/bin/bash
bash tools/r04_final3.sh || exit 1
timeout -k 10 400 python3 tools/small_blocks.py > gpurun_out/r04/small_blocks.txt 2>&1; tail -12 gpurun_out/r04/small_blocks.txt
