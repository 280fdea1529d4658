#!/bin/bash
mkdir -p gpurun_out/r04
{
bash tools/ab_bench.sh "C2V0 C2V64 C2V36 C2V100 C2V79 C2V111 C2V0 C2V100" "mid" 524288
bash tools/ab_bench.sh "C2V0 C2V79 C2V111 C2V0" "max+e8e9 min" 524288
} > gpurun_out/r04/ab7.log 2>&1
cat gpurun_out/r04/ab7.log
