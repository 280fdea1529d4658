#!/bin/bash
mkdir -p gpurun_out/r04
bash tools/prof_stages.sh mid 524288 C2_PROF_MASK0x6000v15 > gpurun_out/r04/prof_vm3_mid.log 2>&1; cat gpurun_out/r04/prof_vm3_mid.log
bash tools/prof_stages.sh max+e8e9 524288 C2_PROF_MASK0x6000v15 > gpurun_out/r04/prof_vm3_max.log 2>&1; cat gpurun_out/r04/prof_vm3_max.log
