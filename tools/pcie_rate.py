#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (never the bench `value`): wall time of
zpaqhip_decompress() from a compressed stream in host memory to plaintext in host memory, for the
headline workload (256 x 4 MiB, level-1 model).  Prints one JSON line.  Usage: tools/pcie_rate.py [--blocks N --block-bytes B --model M]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=256)
ap.add_argument("--block-bytes", type=int, default=4 << 20)
ap.add_argument("--model", default="l1")
ap.add_argument("--kind", default="T")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()

import zpaqsharp_amd as z  # noqa: E402
from zpaqsharp_amd import models, synth  # noqa: E402

stream, _ = synth.stream(models.get(a.model), a.kind, a.blocks, a.block_bytes, threads=16)
ctx = z.Context(0)
plain = a.blocks * a.block_bytes
res = {}
for label, opt in (("no_sha1", dict(verify_sha1=False)), ("verify_sha1", dict(verify_sha1=True))):
    ctx.decompress(stream, out_cap=plain, **opt)                       # warm-up (allocations, table upload)
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        out = ctx.decompress(stream, out_cap=plain, **opt)
        ts.append(time.perf_counter() - t0)
    assert out.size == plain
    st = ctx.stats()
    res[label] = {"MB_per_s": plain / min(ts) / 1e6, "seconds": min(ts), "kernel_ms": st.kernel_ms}
ok = all(np.array_equal(out[b * a.block_bytes:(b + 1) * a.block_bytes], synth.plain(a.kind, b, a.block_bytes)) for b in (0, a.blocks - 1))
print(json.dumps({"workload": f"{a.blocks} x {a.block_bytes >> 20} MiB, model {a.model}, plaintext {a.kind}", "coded_bytes": int(stream.size),
                  "plain_bytes": plain, "bit_exact_sample": bool(ok), "host_to_host": res}))
