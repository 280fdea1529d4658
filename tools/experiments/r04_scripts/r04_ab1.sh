#!/bin/bash
mkdir -p gpurun_out/r04
{
bash tools/ab_bench.sh "C2V0 C2V1 C2V3 C2V5 C2V7 C2V0 C2V7" "mid" 524288
bash tools/ab_bench.sh "C2V0 C2V7 C2V3 C2V0 C2V7" "max+e8e9 min" 524288
} > gpurun_out/r04/ab1.log 2>&1
cat gpurun_out/r04/ab1.log
bash tools/prof_stages.sh mid 524288 C2_PROF_MASK0x6000 > gpurun_out/r04/prof_vm_mid.log 2>&1; cat gpurun_out/r04/prof_vm_mid.log
bash tools/prof_stages.sh max+e8e9 524288 C2_PROF_MASK0x6000 > gpurun_out/r04/prof_vm_max.log 2>&1; cat gpurun_out/r04/prof_vm_max.log
cp build/ab/libC2V7.so zpaqsharp_amd/libzpaqhip.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputests1.log 2>&1; tail -5 gpurun_out/r04/gputests1.log
