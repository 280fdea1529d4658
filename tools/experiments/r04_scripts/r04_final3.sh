#!/bin/bash
# L1 after the shadow work: parity + fuzz + T/X/R, stage tables (byte, miss), the hash-guarded profile set, then the whole GPU suite
mkdir -p gpurun_out/r04
bash tools/r04_l1.sh || exit 1
cp gpurun_out/r04/l1_txr2.txt gpurun_out/r04/l1_txr_256x1MiB.txt
bash tools/prof_l1_stages.sh gpurun_out/r04/stages_l1_byte.txt T || exit 1
bash tools/prof_l1_miss.sh gpurun_out/r04/stages_l1_miss.txt || exit 1
bash tools/r04_profiles.sh l1 > gpurun_out/r04/profiles_l1.log 2>&1 || { tail -5 gpurun_out/r04/profiles_l1.log; exit 1; }
tail -4 gpurun_out/r04/profiles_l1.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gpu_tests_final.log 2>&1; tail -3 gpurun_out/r04/gpu_tests_final.log
