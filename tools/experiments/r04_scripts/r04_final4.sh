#!/bin/bash
# the default command under rocprofv3 (kernel stats), then the line itself
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/default_trace -- python3 bench.py > gpurun_out/r04/default_bench_traced.json 2> gpurun_out/r04/default_trace.err || tail -3 gpurun_out/r04/default_trace.err
cp gpurun_out/r04/default_trace/*/*_kernel_stats.csv gpurun_out/r04/default_kernel_stats.csv; head -8 gpurun_out/r04/default_kernel_stats.csv
bash tools/r04_bench.sh
