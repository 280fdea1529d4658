#!/bin/bash
mkdir -p gpurun_out/r04
tools/ubench/mix_bench > gpurun_out/r04/mix_bench.txt 2>&1; cat gpurun_out/r04/mix_bench.txt
{
bash tools/ab_bench.sh "C2V0 C2V15 C2V31 C2V47 C2V63 C2V0 C2V63" "mid" 524288
bash tools/ab_bench.sh "C2V0 C2V15 C2V31 C2V63 C2V0" "max+e8e9 min" 524288
} > gpurun_out/r04/ab3.log 2>&1
cat gpurun_out/r04/ab3.log
bash tools/prof_stages.sh mid 524288 C2_PROF_MASK0x6000v0 C2_PROF_MASK0x6000v63 > gpurun_out/r04/prof_vm2_mid.log 2>&1; cat gpurun_out/r04/prof_vm2_mid.log
