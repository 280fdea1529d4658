#!/usr/bin/env python3
"""Folds the rocprofv3 counter CSVs of tools/profile_bench.sh into pmc_<model>_<blocks>x<KiB>KiB.json:
per-launch FETCH_SIZE / WRITE_SIZE (KB, as reported) of the decode kernel, the kernel name and the hash of the
kernel sources they were taken with (bench.py only trusts a summary whose hash matches the sources it runs)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    out, model, nb, bs = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    res = {"src_hash": bench.source_hash(model), "model": model, "blocks": nb, "block_bytes": bs}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals, name = [], None
        for f in glob.glob(os.path.join(out, f"pmc_{c}", "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == c and "zh_decode" in r.get("Kernel_Name", ""):
                    vals.append(float(r["Counter_Value"]))
                    name = r["Kernel_Name"]
        if not vals:
            raise SystemExit(f"no {c} rows for a zh_decode kernel under {out}")
        res[c + "_KB"] = sum(vals) / len(vals)
        res["kernel"] = name
        res["launches_seen"] = len(vals)
    path = os.path.join(out, f"pmc_{model.replace('+', '_')}_{nb}x{bs >> 10}KiB.json")
    json.dump(res, open(path, "w"), indent=1)
    print(open(path).read())


if __name__ == "__main__":
    main()
