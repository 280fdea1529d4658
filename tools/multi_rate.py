"""On the GPU box: what zpaqhip_decompress_multi's work queue costs beside the single-context pipeline, host to host.

  python tools/multi_rate.py [--model l1] [--blocks 256] [--kib 4096]

Decodes the same stream with Context.decompress (one context, batch pipeline), with zpaqhip_decompress_multi on [0]
(one device thread pulling chunks of 256 blocks) and on [0, 0] (two device threads sharing this box's one GPU, chunks of
64), checks every block and prints the rates."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="l1")
    ap.add_argument("--blocks", type=int, default=256)
    ap.add_argument("--kib", type=int, default=4096)
    a = ap.parse_args()
    import zpaqsharp_amd as z
    from zpaqsharp_amd import synth
    bs = a.kib << 10
    s, _ = synth.stream(a.model, "T", a.blocks, bs)
    ctx = z.Context(0)

    def check(out):
        return out.size == bs * a.blocks and all(np.array_equal(out[b * bs:(b + 1) * bs], synth.plain("T", b, bs)) for b in range(a.blocks))
    runs = (("one context, batch pipeline", lambda: ctx.decompress(s, out_cap=bs * a.blocks)),
            ("multi [0], queue_blocks 256", lambda: z.decompress_multi([0], s, out_cap=bs * a.blocks)),
            ("multi [0, 0], queue_blocks 64", lambda: z.decompress_multi([0, 0], s, out_cap=bs * a.blocks, queue_blocks=64)))
    def via_callbacks():                                      # zpaqhip_decompress_cb: Reader / Writer shaped callbacks
        mv, pos, chunks = memoryview(s), [0], []

        def rd(n):
            b = mv[pos[0]:pos[0] + n]
            pos[0] += len(b)
            return bytes(b)
        ctx.decompress_cb(rd, chunks.append)
        return np.frombuffer(b"".join(chunks), np.uint8)
    runs = runs + (("one context, Reader / Writer callbacks", via_callbacks),)
    for name, fn in runs:
        fn()                                                  # warm-up
        t0 = time.time()
        out = fn()
        dt = time.time() - t0
        print(f"{a.model} {a.blocks} x {a.kib} KiB  {name:40s} {bs * a.blocks / dt / 1e6:8.1f} MB/s host to host  exact={check(out)}", flush=True)


if __name__ == "__main__":
    main()
