#!/bin/bash
# Builds variants of ONE kernel file for a same-box A/B (tools/ab_bench.sh): tools/build_variants.sh <file.hip> <macro> <values...>
# (EXTRA="-DOTHER=1" adds flags to every variant; TAG=x names the outputs lib<macro><value>x.so)
# -> build/ab/lib<macro><value>.so (the other objects are the ones already built in zpaqsharp_amd/csrc).
set -e
SRC=$1; MAC=$2; shift 2
cd "$(dirname "$0")/../zpaqsharp_amd/csrc"
mkdir -p ../../build/ab
make -s -j8
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$" | grep -v zh_chain3.o)
for v in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -D$MAC=$v $EXTRA --offload-arch=gfx950 -c $SRC -o /tmp/${SRC%.hip}_$MAC$v$TAG.o 2>/dev/null &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/ab/lib$MAC$v$TAG.so $OBJS /tmp/${SRC%.hip}_$MAC$v$TAG.o && echo built $MAC$v$TAG ) &
done
wait
