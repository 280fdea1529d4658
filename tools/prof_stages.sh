#!/bin/bash
# On the GPU box: stage cycles of the *_prof kernels (ZPAQHIP_PROF=1: in-kernel s_memtime stamps summed per block into L.debug)
# for one model.  Usage: tools/prof_stages.sh <model> <block-bytes> [lib tags in build/ab ...]; prints, per tag, the sums and
# cycles per decoded bit.  The library in place at the end is the one that was there at the start.
MODEL=${1:-mid}; BS=${2:-1048576}; shift 2
cp zpaqsharp_amd/libzpaqhip.so /tmp/lib_keep.so
for v in "$@"; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so || exit 1
  ZPAQHIP_PROF=1 timeout -k 10 400 python3 bench.py --model $MODEL --blocks 256 --block-bytes $BS --steps 1 --warmup 0 --no-extras --no-cpu-baseline --cache-dir /tmp/zc > /tmp/prof.json 2> /tmp/prof.err || { tail -3 /tmp/prof.err; exit 1; }
  python3 - "$v" "$MODEL" "$BS" <<'PY'
import json, sys
tag, model, bs = sys.argv[1], sys.argv[2], int(sys.argv[3])
d = json.loads(open('/tmp/prof.json').read().strip().splitlines()[-1])
line = [l for l in open('/tmp/prof.err') if l.startswith('ZPAQHIP_PROF cycles:')][-1]
c = [int(x) for x in line.split(':')[1].split()]
bits = 256 * bs * 8
print(tag, model, 'MB/s', round(d['value'], 2), 'kernel_ms', round(d['roofline']['kernel_ms'], 1), 'exact', d['bit_exact'])
print(tag, 'cycles per bit by stage:', [round(x / bits, 1) for x in c], 'sum', round(sum(c) / bits, 1))
PY
done
cp /tmp/lib_keep.so zpaqsharp_amd/libzpaqhip.so
