#!/bin/bash
mkdir -p gpurun_out/r04
{
bash tools/ab_bench.sh "C2V0 C2V8 C2V1 C2V9 C2V11 C2V15 C2V0 C2V15" "mid" 524288
bash tools/ab_bench.sh "C2V0 C2V8 C2V15 C2V11 C2V0" "max+e8e9 min" 524288
} > gpurun_out/r04/ab2.log 2>&1
cat gpurun_out/r04/ab2.log
