"""Development check of the nibble-at-a-time chain kernels (zh_nibble.hip) against the bit-at-a-time ones (zh_chain2.hip), the
oracle and the plaintext, then a same-box rate comparison.  On the GPU box:
    python tools/dev_nibble.py [check|rate|all] [models...] [--kib N] [--blocks N]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpaqsharp_amd as z
from zpaqsharp_amd import synth, models

NEW = int(os.environ.get("NB_KERNEL", "0"))       # opts.kernel of the form under test
OLD = int(os.environ.get("NB_OLD", "9"))          # ... and of the form it is compared with


def check(ctx, mods):
    import oracle
    from tests import util
    rng = np.random.default_rng(5)
    bad = 0
    for model in mods:
        cases = {
            "text64k": util.text(65536),
            "text3": util.text(3),
            "empty": b"",
            "one": b"a",
            "zeros": bytes(30000),
            "runs": np.repeat(rng.integers(0, 256, 600, dtype=np.uint8), 50).tobytes(),
            "random": rng.integers(0, 256, 40000, dtype=np.uint8).tobytes(),
            "few": rng.integers(0, 4, 40000, dtype=np.uint8).tobytes(),
            "period": (b"abcdefghijklmnopq" * 4000)[:50000],
            "x86": util.x86ish(50000),
            "text300k": util.text(300000, seed=9),
        }
        for name, data in cases.items():
            s = util.block(model, data)
            try:
                got = ctx.decompress(s, verify_sha1=True, kernel=NEW, out_cap=len(data) + 16)
            except Exception as e:
                print(f"FAIL {model} {name}: {e}", flush=True)
                bad += 1
                continue
            ok = got.tobytes() == data
            if not ok:
                ref = ctx.decompress(s, verify_sha1=True, kernel=OLD, out_cap=len(data) + 16).tobytes()
                g = got.tobytes()
                n = min(len(g), len(data))
                first = next((i for i in range(n) if g[i] != data[i]), n)
                print(f"FAIL {model} {name}: len {len(g)} / {len(data)}, first difference at {first}, old kernel ok={ref == data}", flush=True)
                bad += 1
            else:
                print(f"ok   {model} {name} ({len(data)} bytes)", flush=True)
        # several blocks, several segments
        parts = [util.text(20000, seed=3), b"", util.text(7777, seed=4), rng.integers(0, 256, 9000, dtype=np.uint8).tobytes()]
        c = oracle.Compressor(400000)
        c.write_tag(); c.start_block(models.get(model).header)
        for i, p_ in enumerate(parts):
            c.start_segment(b"f%d" % i, str(len(p_)).encode())
            if i == 0:
                c.post_process(models.get(model).pcomp)
            c.compress(p_); c.end_segment(oracle.sha1(p_))
        c.end_block()
        ms = c.getvalue(); c.close()
        try:
            got = ctx.decompress(ms, verify_sha1=True, kernel=NEW, out_cap=sum(map(len, parts)) + 16).tobytes()
        except Exception as e:
            got = repr(e).encode()
        ok = got == b"".join(parts)
        print(("ok  " if ok else "FAIL") + f" {model} multi-segment", flush=True)
        bad += not ok
        s, _ = synth.stream(model, "T", 300, 20000, threads=8)
        want = np.concatenate([synth.plain("T", b, 20000) for b in range(300)])
        try:
            got = ctx.decompress(s, verify_sha1=True, kernel=NEW, out_cap=want.size)
            ok = np.array_equal(got, want)
        except Exception as e:
            print(e); ok = False
        print(("ok  " if ok else "FAIL") + f" {model} 300 blocks x 20000", flush=True)
        bad += not ok
    return bad


def rate(ctx, mods, nb, kib, kinds="T"):
    bs = kib << 10
    for model in mods:
        for kind in kinds:
            s, _ = synth.stream(model, kind, nb, bs, threads=16)
            for kern in (OLD, NEW, OLD, NEW):
                out = ctx.decompress(s, out_cap=nb * bs, kernel=kern)
                st = ctx.stats()
                ok = all(np.array_equal(out[b * bs:(b + 1) * bs], synth.plain(kind, b, bs)) for b in range(0, nb, max(1, nb // 16)))
                print(f"{model} {kind} {nb} x {kib} KiB kernel={kern}: kernel {st.kernel_ms:9.1f} ms = {nb * bs / st.kernel_ms / 1e3:7.2f} MB/s "
                      f"({st.kernel_ms * 1e-3 * 2.4e9 / bs:7.0f} cycles/byte), exact={ok}", flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    what = args[0] if args else "all"
    mods = args[1:] or ["min", "mid"]
    kib = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--kib=")), 256))
    nb = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--blocks=")), 256))
    kinds = next((a.split("=")[1] for a in sys.argv if a.startswith("--kinds=")), "T")
    ctx = z.Context(0)
    bad = 0
    if what in ("check", "all"):
        bad = check(ctx, mods)
    if what in ("rate", "all") and not bad:
        rate(ctx, mods, nb, kib, kinds)
    sys.exit(1 if bad else 0)
