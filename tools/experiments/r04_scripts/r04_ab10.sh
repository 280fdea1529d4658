#!/bin/bash
mkdir -p gpurun_out/r04
{
bash tools/ab_bench.sh "C2V239 C2V495 C2V239 C2V495" "mid min" 524288
} > gpurun_out/r04/ab10.log 2>&1
cat gpurun_out/r04/ab10.log
cp build/ab/libC2V239.so zpaqsharp_amd/libzpaqhip.so
bash tools/r04_bench.sh
