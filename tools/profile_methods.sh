#!/bin/bash
# On the GPU box: rocprofv3 kernel trace and FETCH_SIZE / WRITE_SIZE PMC passes (separate runs) over the store kernel
# (zh_decode_store) for the reference's unmodelled methods, 256 distinct 4 MiB blocks each -> gpurun_out/methods/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/methods; mkdir -p $OUT
for M in "x2,1,4,0,3,22" "x2,2,12,0,7,22" "x3,3"; do
  T=$(echo $M | tr ',' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$T -- python3 tools/method_rate.py --methods "$M" --json $OUT/rate_$T.json > $OUT/trace_$T.log 2>&1 || tail -3 $OUT/trace_$T.log
  cp $OUT/trace_$T/*/*_kernel_stats.csv $OUT/kernel_stats_$T.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_$T -- python3 tools/method_rate.py --methods "$M" > $OUT/pmc_${C}_$T.log 2>&1 || tail -3 $OUT/pmc_${C}_$T.log
  done
  python3 - $OUT $T "$M" <<'PY'
import csv, glob, json, sys
out, t, m = sys.argv[1:4]
res = {"method": m, "blocks": 256, "block_bytes": 4 << 20}
for r in csv.DictReader(open(f"{out}/kernel_stats_{t}.csv")):
    if "zh_decode_store" in r["Name"]:
        res["kernel"] = r["Name"]; res["calls"] = int(r["Calls"]); res["avg_ns"] = float(r["AverageNs"])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = [float(r["Counter_Value"]) for f in glob.glob(f"{out}/pmc_{c}_{t}/*/*counter_collection.csv") for r in csv.DictReader(open(f))
            if r.get("Counter_Name") == c and "zh_decode_store" in r.get("Kernel_Name", "")]
    res[c + "_KB_per_launch"] = sum(vals) / len(vals) if vals else None
rate = json.load(open(f"{out}/rate_{t}.json"))[0]
res["coded_bytes"] = rate["coded_bytes"]; res["kernel_MBps_events"] = rate["kernel_MBps"]; res["bit_exact"] = rate["bit_exact"]
json.dump(res, open(f"{out}/store_{t}.json", "w"), indent=1)
print(json.dumps(res))
PY
done
