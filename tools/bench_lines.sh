#!/bin/bash
# On the GPU box: the bench line of every BASELINE-shaped workload (256 x 4 MiB) with the committed PMC summaries and
# instruction counts of these sources in place -> gpurun_out/lines/<model>_256x4MiB_bench.json
mkdir -p gpurun_out/lines
for m in "$@"; do
  t=$(echo $m | tr '+' '_')
  python3 bench.py --model $m --blocks 256 --block-bytes 4194304 --no-extras --cache-dir /tmp/zc > gpurun_out/lines/${t}_256x4MiB_bench.json 2> gpurun_out/lines/$t.err || tail -3 gpurun_out/lines/$t.err
  python3 -c "
import json
d=json.load(open('gpurun_out/lines/${t}_256x4MiB_bench.json')); r=d['roofline']
print('$m', round(d['value'],2), d['bit_exact'], 'kernel_ms', round(r['kernel_ms'],1), 'traffic', r['traffic'], 'instr', r.get('instr_per_byte'), 'cycles/byte', r.get('cycles_per_byte'))"
done
