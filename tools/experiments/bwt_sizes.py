"""How the modelled BWT method (x0,3ci1 / x4,3ci1) scales with the block size: kernel ms per configuration."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zpaqsharp_amd as z
from zpaqsharp_amd import synth
from tools import methods
ctx = z.Context(0)
for mt, nb, kib in (("x0,3ci1", 64, 64), ("x0,3ci1", 64, 128), ("x0,3ci1", 64, 256), ("x0,3ci1", 16, 512), ("x4,3ci1", 16, 1024)):
    model, margs = methods.model_of(mt)
    bs = kib << 10
    s, _ = synth.method_stream(model, margs, "T", nb, bs, threads=16)
    for kern in (0, 4):
        t0 = time.time()
        out = ctx.decompress(s, out_cap=nb * bs, kernel=kern)
        st = ctx.stats()
        ok = all(np.array_equal(out[b * bs:(b + 1) * bs], synth.plain("T", b, bs)) for b in range(nb))
        print(f"{mt} {nb} x {kib} KiB kernel={kern}: kernel {st.kernel_ms:9.1f} ms ({st.kernel_ms * 1e-3 * 2.4e9 / bs:8.0f} cycles/byte/block) exact={ok} wall {time.time() - t0:.1f}s", flush=True)
