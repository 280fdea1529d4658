#!/bin/bash
# On the GPU box: everything profiles/r04/ keeps for one model — the bench line, rocprofv3 kernel stats, FETCH/WRITE PMC
# (tools/profile_bench.sh, BASELINE size 256 x 4 MiB) and the instruction mix (tools/pmc_mix.sh, 256 x 1 MiB).
# Usage: tools/r04_profiles.sh <model> [plaintext kind]; results under gpurun_out/r04_<model>/ and gpurun_out/instmix_<model>/
M=$1; K=${2:-}; TAG=$(echo $M$K | tr '+' '_')
export ROUND=r04
if [ -n "$K" ]; then bash tools/profile_bench.sh r04_$TAG $M 256 1048576 --kind $K || exit 1
else bash tools/profile_bench.sh r04_$TAG $M 256 4194304 || exit 1; bash tools/pmc_mix.sh $M 1048576; fi
