import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import zpaqsharp_amd as z
from zpaqsharp_amd import synth
ctx = z.Context(0)
for nb, kib in ((256, 1024), (4096, 64), (16384, 16), (65536, 4)):
    bs = kib << 10
    s, _ = synth.stream("l1", "T", nb, bs)
    ctx.decompress(s, out_cap=nb * bs)
    t0 = time.time(); out = ctx.decompress(s, out_cap=nb * bs); dt = time.time() - t0
    st = ctx.stats()
    ok = all(np.array_equal(out[b * bs:(b + 1) * bs], synth.plain("T", b, bs)) for b in range(0, nb, max(1, nb // 64)))
    print(f"l1 {nb} x {kib} KiB: {nb * bs / dt / 1e6:7.1f} MB/s host to host, kernel {st.kernel_ms:7.1f} ms of {dt * 1e3:7.1f} ms, launches {st.launches}, exact={ok}", flush=True)
