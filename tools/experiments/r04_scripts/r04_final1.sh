#!/bin/bash
mkdir -p gpurun_out/r04
bash tools/ab_bench.sh "C2V1007 C2V999 C2V1007 C2V999" "mid max+e8e9 min" 524288 > gpurun_out/r04/ab12.log 2>&1; cat gpurun_out/r04/ab12.log
cp build/ab/libC2V999.so zpaqsharp_amd/libzpaqhip.so
for m in min mid; do bash tools/r04_profiles.sh $m > gpurun_out/r04_prof_$m.log 2>&1; tail -1 gpurun_out/r04_prof_$m.log; done
