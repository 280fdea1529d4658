import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import zpaqsharp_amd as z
from zpaqsharp_amd import synth
ctx = z.Context(0)
nb, bs = 256, 256 << 10
for model, kind in (("max+e8e9", "X"), ("max", "T")):
    s, _ = synth.stream(model, kind, nb, bs, threads=16)
    for it in range(2):
        out = ctx.decompress(s, out_cap=nb * bs, verify_sha1=(model == "max"))
        st = ctx.stats()
        ok = all(np.array_equal(out[b * bs:(b + 1) * bs], synth.plain(kind, b, bs)) for b in range(0, nb, 16))
        print(f"{model} {kind} {nb} x {bs >> 10} KiB: kernel {st.kernel_ms:9.1f} ms = {nb * bs / st.kernel_ms / 1e3:7.2f} MB/s ({st.kernel_ms * 1e-3 * 2.4e9 / bs:7.0f} cycles/byte), exact={ok}", flush=True)
