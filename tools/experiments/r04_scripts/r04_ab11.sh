#!/bin/bash
mkdir -p gpurun_out/r04
{
bash tools/ab_bench.sh "C2V239 C2V751 C2V1007 C2V239 C2V751 C2V1007" "mid min" 524288
bash tools/ab_bench.sh "C2V239 C2V751 C2V239 C2V751" "max+e8e9" 524288
} > gpurun_out/r04/ab11.log 2>&1
cat gpurun_out/r04/ab11.log
