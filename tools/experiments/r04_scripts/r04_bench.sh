#!/bin/bash
# default bench line + a readable digest (GPU box)
mkdir -p gpurun_out/r04
S=$(date +%s)
python bench.py > gpurun_out/r04/default_bench.json 2> gpurun_out/r04/default_bench.err
echo "default bench.py: rc $? in $(( $(date +%s) - S )) s"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/default_bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"],1), d["bit_exact"], "host_to_host", d.get("value_host_to_host"), "frac", d["roofline"]["frac"], "cpu1", d["cpu_baseline"]["value"])
for o in d.get("other_configs",[]): print(" ", o["config"][:60], round(o["value"],1), o["bit_exact"], (o.get("cpu_baseline") or {}).get("value"), (o.get("cpu_all_cores") or {}).get("value"))
print(" cpu_share", d.get("cpu_share")); print(" l1 sweep", d.get("cpu_all_cores_sweep"))
for o in d.get("other_configs",[]):
  if "cpu_all_cores_sweep" in o: print(" mid sweep", o["cpu_all_cores_sweep"])
for m in d.get("method_streams",[]): print(" ", m["method"], round(m["value"]), m.get("kernel_MBps"))
PY
