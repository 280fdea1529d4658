"""Decode rate of streams written with the reference's method strings (post-processor paths), on the GPU box.

  python tools/method_rate.py [--kib 4096] [--blocks 256] [--methods "x0,1,4,0,3,16;..."] [--json out.json]

Every block is DISTINCT (synthetic plaintext `kind`, block index b), pre-processed by the method's LZ77 / BWT
pre-processor (zpaqgen, the formats of LZBuffer.cs:96-115) and framed as LibZPAQ.compressBlock does.  The stream is
decoded host to host through zpaqhip_decompress with the stored SHA-1 of every segment verified on the device; the
plaintext of every block is compared on the host.  Prints one line per method and, with --json, writes them."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kib", type=int, default=4096)
    ap.add_argument("--blocks", type=int, default=256)
    ap.add_argument("--kind", default="T")
    ap.add_argument("--methods", default="x2,1,4,0,3,22;x2,2,12,0,7,22;x2,3;x2,5,4,0,3,22")
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    import zpaqsharp_amd as z
    from tools import methods
    from zpaqsharp_amd import synth
    ctx = z.Context(0)
    bs = a.kib << 10
    rows = []
    for mt in a.methods.split(";"):
        model, args = methods.model_of(mt)
        t0 = time.time()
        s, offs = synth.method_stream(model, args, a.kind, a.blocks, bs, threads=a.threads)
        tgen = time.time() - t0
        got = ctx.decompress(s, out_cap=bs * a.blocks, verify_sha1=True)      # warm-up (arena allocation) + check
        ok = got.size == bs * a.blocks
        for b in range(a.blocks):
            if not ok:
                break
            ok = np.array_equal(got[b * bs:(b + 1) * bs], synth.plain(a.kind, b, bs))
        t0 = time.time()
        ctx.decompress(s, out_cap=bs * a.blocks)
        dt = time.time() - t0
        st = ctx.stats()
        row = {"method": mt, "blocks": a.blocks, "block_bytes": bs, "plaintext": a.kind, "coded_bytes": int(s.size),
               "host_to_host_MBps": bs * a.blocks / dt / 1e6, "kernel_ms": st.kernel_ms, "kernel_MBps": bs * a.blocks / (st.kernel_ms * 1e-3) / 1e6,
               "launches": int(st.launches), "bit_exact": bool(ok), "gen_seconds": tgen}
        rows.append(row)
        print(f"{mt:24s} {a.blocks} x {a.kib} KiB distinct, coded {s.size / 1e6:8.1f} MB: {row['host_to_host_MBps']:8.1f} MB/s host-to-host, "
              f"kernel {st.kernel_ms:8.1f} ms = {row['kernel_MBps']:8.1f} MB/s, launches {st.launches}, exact={ok}, gen {tgen:.1f} s", flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
