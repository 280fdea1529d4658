#!/usr/bin/env python3
"""Decode rate of streams written with the reference's method strings (LZ77 / BWT post-processors on the device VM):
one block of --kib KiB text-like plaintext per method, replicated --blocks times (identical blocks decode
independently, so the figure is the throughput of that post-processor path).  Run on the GPU box."""
import argparse
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kib", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=256)
    ap.add_argument("--methods", default="x0,1,4,0,3,16;x0,2,12,0,7,16;x0,3ci1;x0,5,4,0,3,16")
    a = ap.parse_args()
    import zpaqsharp_amd as z
    from tools import methods
    from zpaqsharp_amd import synth
    ctx = z.Context(0)
    plain = synth.plain("T", 7, a.kib << 10).tobytes()
    for mt in a.methods.split(";"):
        t0 = time.time()
        blk = methods.compress_block(mt, plain)
        tgen = time.time() - t0
        s = blk * a.blocks
        got = ctx.decompress(s)                       # warm-up (arena allocation)
        ok = got.size == len(plain) * a.blocks and got[:len(plain)].tobytes() == plain and got[-len(plain):].tobytes() == plain
        t0 = time.time()
        ctx.decompress(s)
        dt = time.time() - t0
        st = ctx.stats()
        print(f"{mt:28s} block {len(blk):7d} B coded, {a.blocks} x {a.kib} KiB: {len(plain) * a.blocks / dt / 1e6:8.1f} MB/s host-to-host "
              f"(kernel {st.kernel_ms:8.1f} ms, kind {st.kernel_kind}), exact={ok}, encode {tgen:.1f} s", flush=True)


if __name__ == "__main__":
    main()
