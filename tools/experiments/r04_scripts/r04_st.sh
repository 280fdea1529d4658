#!/bin/bash
mkdir -p gpurun_out/r04
cp zpaqsharp_amd/libzpaqhip.so /tmp/keep.so; cp build/ab/libST_PROF1.so zpaqsharp_amd/libzpaqhip.so
ZPAQHIP_PROF=1 timeout -k 10 300 python tools/method_rate.py --methods "x2,1,4,0,3,22" > gpurun_out/r04/st_prof.txt 2> gpurun_out/r04/st_prof.err
cp /tmp/keep.so zpaqsharp_amd/libzpaqhip.so
cat gpurun_out/r04/st_prof.txt; grep "ZPAQHIP_PROF" gpurun_out/r04/st_prof.err | tail -2
