#!/bin/bash
# On the GPU box: where a byte of the single-CM fast loop spends its cycles (text, 256 x 1 MiB): four builds of the diagnostic
# kernel, one interval each (-DZH_L1_STAGE=1..4, zh_cm_fast.h), made with
#   for n in 1 2 3 4; do tools/build_variants.sh zh_cm.hip ZH_L1_STAGE $n; done
OUT=${1:-gpurun_out/r04/stages_l1_byte.txt}
KIND=${2:-T}
cp zpaqsharp_amd/libzpaqhip.so /tmp/lib_keep.so
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for n in 1 2 3 4; do
  cp build/ab/libZH_L1_STAGE$n.so zpaqsharp_amd/libzpaqhip.so || exit 1
  ZPAQHIP_PROF=1 timeout -k 10 300 python3 bench.py --model l1 --kind $KIND --blocks 256 --block-bytes 1048576 --steps 1 --warmup 0 --no-extras --no-cpu-baseline --cache-dir /tmp/zc > /tmp/ps.json 2> /tmp/ps.err || { tail -3 /tmp/ps.err; cp /tmp/lib_keep.so zpaqsharp_amd/libzpaqhip.so; exit 1; }
  python3 - "$n" "$KIND" >> "$OUT" <<'PY'
import json, sys
n, kind = int(sys.argv[1]), sys.argv[2]
d = json.loads(open('/tmp/ps.json').read().strip().splitlines()[-1])
c = [int(x) for x in [l for l in open('/tmp/ps.err') if l.startswith('ZPAQHIP_PROF cycles:')][-1].split(':')[1].split()]
what = {1: "byte start -> lag test passed (window lookup and lag test: two v_cmp, s_ff1, s_and_b64)",
        2: "-> probabilities back (LDS addresses, three ds_read, chunk test, EOS flag, s_waitcnt)",
        3: "-> eight bit steps and the nibble switch",
        4: "-> epilogue (byte, message, t, h[0]) and loop branch, to the next byte's start"}[n]
print(f"stage {n}: {c[11] / max(1, c[3]):7.1f} cycles per byte   {what}   [{kind}, {d['value']:.0f} MB/s in this build, whole loop {c[2] / max(1, c[3]):.0f} cycles per byte, bit_exact {d['bit_exact']}]")
PY
done
cp /tmp/lib_keep.so zpaqsharp_amd/libzpaqhip.so
cat "$OUT"
