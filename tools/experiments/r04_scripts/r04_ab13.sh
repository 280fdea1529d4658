#!/bin/bash
mkdir -p gpurun_out/r04
{
bash tools/ab_bench.sh "Sbase Sminsmem Sbase Sminsmem" "min" 524288
bash tools/ab_bench.sh "Sbase Smaxsmem Smaxbal Sbase Smaxsmem Smaxbal" "max+e8e9" 524288
} > gpurun_out/r04/ab13.log 2>&1
cat gpurun_out/r04/ab13.log
