#!/bin/bash
# On the GPU box: the stage table of a window miss of the single-CM kernel (zh_decode_cm_prof, ZPAQHIP_PROF=1) on the
# x86-like and random plaintexts, 256 x 1 MiB.  Usage: tools/prof_l1_miss.sh [out file]
OUT=${1:-gpurun_out/r04/stages_l1_miss.txt}
: > "$OUT"
for K in T X R; do
  ZPAQHIP_PROF=1 timeout -k 10 300 python3 bench.py --model l1 --kind $K --blocks 256 --block-bytes 1048576 --steps 1 --warmup 0 --no-extras --no-cpu-baseline --cache-dir /tmp/zc > /tmp/pl1.json 2> /tmp/pl1.err || { tail -3 /tmp/pl1.err; exit 1; }
  python3 - "$K" >> "$OUT" <<'PY'
import json, sys
k = sys.argv[1]
d = json.loads(open('/tmp/pl1.json').read().strip().splitlines()[-1])
c = [int(x) for x in [l for l in open('/tmp/pl1.err') if l.startswith('ZPAQHIP_PROF cycles:')][-1].split(':')[1].split()]
nbytes = 256 * 1048576
miss = max(1, c[9])
print(f"plaintext {k}: {d['value']:.1f} MB/s (diagnostic build), bit_exact {d['bit_exact']}; swaps {c[9]} = {100.0 * c[9] / nbytes:.1f} % of bytes "
      f"({c[10]} of them asked for by the C++ body); cycles per byte in the fast loop {c[2] / nbytes:.0f}; waits for wave B in the fast loop "
      f"{c[12]} ({c[12] / nbytes:.2f} per byte), {c[11] / max(1, c[12]):.0f} cycles each")
print(f"  per swap, wave C: request seen -> loads and write-back issued {c[13] / miss:.0f}; -> window data back {c[14] / miss:.0f}; "
      f"-> installed and reported {c[15] / miss:.0f}   (wave A, when the C++ body asks: loop left -> request published {c[8] / max(1, c[10]):.0f})")
PY
done
cat "$OUT"
