"""Development check of the modelled method strings (LibZPAQ.makeConfig models) on the GPU: every stream is decoded by the default
kernel choice (zh_nibble.hip for the models of min's / mid's shape) and by the lane-per-component kernel (opts.kernel = 4),
both compared with the plaintext; then a rate line per method.  On the GPU box:
    python tools/dev_methods.py [check|rate|all] [--kib=N] [--blocks=N]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpaqsharp_amd as z
from zpaqsharp_amd import synth
from tools import methods

METHODS = ["x0,0ci1,1,1,1,2awm", "x4,4ci1,1,1,1,2awm", "x0,0ci1,1,1,1,2am", "x0,4ci1,1,1,1,2am", "x0,3ci1", "x4,0ci1,1,1,1,2am", "x6,4ci1,1,1,1,2am", "x4,3ci1", "x0,7ci1", "x0,2,12,0,7,21,1c0,0,511i2", "x4,6,12,0,7,25,1c0,0,511i2", "x0,2,5,0,7,21,1c0,0,511", "x4,6,5,0,7,25,1c0,0,511"]


def check(ctx):
    bad = 0
    for mt in METHODS:
        model, margs = methods.model_of(mt)
        for kind, nb, bs in (("T", 40, 30000), ("X", 12, 70001), ("R", 6, 5000), ("T", 3, 1), ("T", 2, 300000)):
            s, _ = synth.method_stream(model, margs, kind, nb, bs, threads=8)
            want = np.concatenate([synth.plain(kind, b, bs) for b in range(nb)])
            res = {}
            for kern in (0, 4):
                try:
                    got = ctx.decompress(s, out_cap=nb * bs + 64, verify_sha1=True, kernel=kern)
                    res[kern] = (np.array_equal(got, want), int(ctx.stats().kernel_kind), float(ctx.stats().kernel_ms))
                except Exception as e:
                    res[kern] = (False, -1, repr(e)[:80])
            ok = res[0][0] and res[4][0]
            bad += not ok
            print(("ok  " if ok else "FAIL") + f" {mt:22s} {kind} {nb:3d} x {bs:6d}: default {res[0]}, lane-per-component {res[4]}", flush=True)
    return bad


def rate(ctx, nb, kib):
    bs = kib << 10
    for mt in METHODS[:5] + METHODS[-4:]:
        model, margs = methods.model_of(mt)
        t0 = time.time()
        s, _ = synth.method_stream(model, margs, "T", nb, bs, threads=16)
        tg = time.time() - t0
        for kern in (0, 4) if kib <= 64 else (0,):
            out = ctx.decompress(s, out_cap=nb * bs, kernel=kern)
            st = ctx.stats()
            ok = all(np.array_equal(out[b * bs:(b + 1) * bs], synth.plain("T", b, bs)) for b in range(0, nb, max(1, nb // 16)))
            print(f"{mt:22s} {nb} x {kib} KiB kernel={kern}: kernel {st.kernel_ms:9.1f} ms = {nb * bs / st.kernel_ms / 1e3:7.2f} MB/s "
                  f"({st.kernel_ms * 1e-3 * 2.4e9 / bs:7.0f} cycles/byte), exact={ok}, stream written in {tg:.0f} s", flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    what = args[0] if args else "all"
    kib = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--kib=")), 256))
    nb = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--blocks=")), 256))
    ctx = z.Context(0)
    bad = 0
    if what in ("check", "all"):
        bad = check(ctx)
    if what in ("rate", "all") and not bad:
        rate(ctx, nb, kib)
    sys.exit(1 if bad else 0)
