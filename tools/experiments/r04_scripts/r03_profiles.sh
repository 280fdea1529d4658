#!/bin/bash
# On the GPU box: everything profiles/r03/ keeps for one model — the bench line, rocprofv3 kernel stats, FETCH/WRITE PMC
# (tools/profile_bench.sh, BASELINE size 256 x 4 MiB) and the instruction mix (tools/pmc_mix.sh, 256 x 1 MiB).
# Usage: tools/r03_profiles.sh <model> ; results under gpurun_out/r03_<model>/ and gpurun_out/instmix_<model>/
M=$1; TAG=$(echo $M | tr '+' '_')
bash tools/profile_bench.sh r03_$TAG $M 256 4194304 || exit 1
bash tools/pmc_mix.sh $M 1048576
