"""The method models at BASELINE size (256 x 4 MiB): kernel rate of the host's kernel choice."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zpaqsharp_amd as z
from zpaqsharp_amd import synth
from tools import methods
ctx = z.Context(0)
nb, bs = 256, 4 << 20
for mt in ("x2,0ci1,1,1,1,2awm", "x2,2,12,0,7,23,1c0,0,511i2", "x2,3ci1", "x2,4ci1,1,1,1,2am"):
    model, margs = methods.model_of(mt)
    t0 = time.time()
    s, _ = synth.method_stream(model, margs, "T", nb, bs, threads=16)
    print(f"{mt}: stream written in {time.time() - t0:.0f} s ({s.size / 1e6:.0f} MB)", flush=True)
    out = ctx.decompress(s, out_cap=nb * bs)
    st = ctx.stats()
    ok = all(np.array_equal(out[b * bs:(b + 1) * bs], synth.plain("T", b, bs)) for b in range(0, nb, 8))
    print(f"{mt} {nb} x {bs >> 10} KiB: kernel {st.kernel_ms:9.1f} ms = {nb * bs / st.kernel_ms / 1e3:7.2f} MB/s ({st.kernel_ms * 1e-3 * 2.4e9 / bs:7.0f} cycles/byte), exact={ok}", flush=True)
    del out, s
