#!/bin/bash
# Same-box A/B of kernel variants (zpaqhip_opts.kernel) on one model: tools/ab_kernels.sh <model> <block-bytes> <kernel ids...>
# 0 auto (two-wave zh_chain2 for min/mid/max), 8 three-wave zh_chain3, 7 three-wave without speculation, 5 lane-per-component
model=$1; bs=$2; shift 2
for k in "$@"; do
  timeout -k 10 400 python bench.py --model $model --blocks 256 --block-bytes $bs --steps 2 --warmup 1 --no-extras --no-cpu-baseline --kernel $k --cache-dir /tmp/zc > gpurun_out/ab_${model}_k$k.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/ab_${model}_k$k.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$model kernel $k: %.2f MB/s bit_exact=%s kernel_ms=%.1f" % (d["value"], d["bit_exact"], d["roofline"]["kernel_ms"]))
PY
done
