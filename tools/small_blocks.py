"""Host-to-host rate of one context on streams of many small single-CM blocks: what a block costs the host, and what more than
one block per CU buys (kernel 2 = one block per workgroup, 256 in flight; 0 = auto: two per workgroup beyond 256 blocks).
On the GPU box: python tools/small_blocks.py [> profiles/r04/small_blocks.txt]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpaqsharp_amd as z
from zpaqsharp_amd import synth
ctx = z.Context(0)
for kind in "TX":
    for nb, kib in ((256, 1024), (512, 1024), (1024, 1024), (4096, 64), (65536, 4)):
        if kind == "X" and nb > 1024:
            continue
        bs = kib << 10
        s, _ = synth.stream("l1", kind, nb, bs)
        for kern in (2, 0):
            ctx.decompress(s, out_cap=nb * bs, kernel=kern)
            t0 = time.time(); out = ctx.decompress(s, out_cap=nb * bs, kernel=kern); dt = time.time() - t0
            st = ctx.stats()
            ok = all(np.array_equal(out[b * bs:(b + 1) * bs], synth.plain(kind, b, bs)) for b in range(0, nb, max(1, nb // 64)))
            print(f"l1 {kind} {nb:6d} x {kib:5d} KiB kernel={kern}: {nb * bs / dt / 1e6:7.1f} MB/s host to host, kernel {st.kernel_ms:7.1f} ms "
                  f"({nb * bs / st.kernel_ms / 1e3:7.1f} MB/s) of {dt * 1e3:7.1f} ms, in flight {st.concurrent}, launches {st.launches}, exact={ok}", flush=True)
