#!/bin/bash
# Same-box A/B of two (or more) builds of libzpaqhip.so: different GPU boxes of the pool differ by ~1 % on one build, two
# builds alternated inside ONE gpurun call agree to ~0.1 %, which is what per-model choices of a few tenths of a percent
# need.  Prepare the builds here (make, then cp zpaqsharp_amd/libzpaqhip.so build/ab/lib<TAG>.so
# for each variant), then on the GPU box:
#   gpurun -- 'bash tools/ab_bench.sh "A B A B" "mid max" 524288 > gpurun_out/ab.log 2>&1'
# The library in place when the script ends is the first tag's.
TAGS=${1:-"A B A B"}; MODELS=${2:-"mid max"}; BS=${3:-524288}
FIRST=$(echo $TAGS | cut -d' ' -f1)
for v in $TAGS; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so || exit 1
  for m in $MODELS; do
    timeout -k 10 300 python3 bench.py --model $m --blocks 256 --block-bytes $BS --no-extras --no-cpu-baseline --cache-dir /tmp/zc > /tmp/ab.json 2>/dev/null || exit 1
    python3 -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('$v', '$m', round(d['value'], 2), d['bit_exact'])"
  done
done
cp build/ab/lib$FIRST.so zpaqsharp_amd/libzpaqhip.so
