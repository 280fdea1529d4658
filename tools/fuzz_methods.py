"""Randomised soak of the modelled method strings on the GPU (zh_nibble.hip's paths that plain text rarely walks: rows handed over
inside a bucket, lzpre matches overlapping their source, literal / match boundaries at chunk edges, tiny and empty blocks, many
blocks of different models in one stream): every decoded stream must equal the plaintext it was written from.
    python tools/fuzz_methods.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpaqsharp_amd as z
from tests import util
from tools import methods

METHODS = ["x0,0ci1,1,1,1,2am", "x0,4ci1,1,1,1,2am", "x4,0ci1,1,1,1,2awm", "x0,4ci1,1,1,1,2awm", "x0,3ci1", "x0,7ci1", "x4,3ci1",
           "x0,2,12,0,7,21,1c0,0,511i2", "x4,6,12,0,7,25,1c0,0,511i2", "x0,2,5,0,7,21,1c0,0,511", "x4,6,5,0,7,25,1c0,0,511", "mid", "min"]


def sample(rng, n, tame=False):
    """tame: no long repetitive inputs (the CPU stream writer's suffix sort / match finder take minutes on them)"""
    kind = int(rng.integers(0, 9))
    if tame and kind in (3, 4, 5, 8) and n > 3000:
        n0, n = n, 3000
        return sample(rng, n, True) + util.text(n0 - n, seed=int(rng.integers(1, 1 << 30)))
    if kind == 0: return util.text(n, seed=int(rng.integers(1, 1 << 30)))
    if kind == 1: return util.x86ish(n, int(rng.integers(1, 1 << 30)))
    if kind == 2: return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 3: return rng.integers(0, int(rng.integers(2, 6)), n, dtype=np.uint8).tobytes()
    if kind == 4:
        rep = int(rng.integers(1, 90))
        return np.repeat(rng.integers(0, 256, n // rep + 1, dtype=np.uint8), rep)[:n].tobytes()
    if kind == 5:
        per = rng.integers(0, 256, int(rng.integers(1, 70)), dtype=np.uint8).tobytes()
        return (per * (n // len(per) + 1))[:n]
    if kind == 6:
        t = bytearray(util.text(n, seed=int(rng.integers(1, 1 << 30))))
        for _ in range(n // 50 + 1 if n else 0): t[int(rng.integers(0, n))] = int(rng.integers(0, 256))
        return bytes(t[:n])
    if kind == 7:
        a = util.text(n // 3 + 1, seed=int(rng.integers(1, 1 << 30)))
        return (a + a[::-1] + a)[:n]
    return bytes(n)


def block(rng, mt, data):
    if mt in ("mid", "min"):
        return util.block(mt, data)
    return methods.compress_block(mt, data)


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    ctx = z.Context(0)
    t0, rounds, nbytes, bad = time.time(), 0, 0, 0
    while time.time() - t0 < secs:
        parts = []
        for _ in range(int(rng.integers(1, 40))):
            mt = METHODS[int(rng.integers(0, len(METHODS)))]
            n = int(rng.choice([0, 1, 2, 39, 40, 41, 255, 256, 257, 1000, 4096, 20000, 70000, 200000], p=[.03, .03, .03, .05, .05, .05, .06, .06, .06, .12, .12, .14, .12, .08]))
            n = max(0, n + int(rng.integers(-3, 4))) if n > 300 else n
            d = sample(rng, n, tame=mt[:2] != "mi" and mt.split(",")[1][0] in "2367")
            parts.append((mt, d))
        s = b"".join(block(rng, mt, d) for mt, d in parts)
        want = b"".join(d for _, d in parts)
        try:
            got = ctx.decompress(s, verify_sha1=True, out_cap=len(want) + 64).tobytes()
            ok = got == want
        except Exception as e:
            ok, got = False, repr(e)[:100].encode()
        rounds += 1; nbytes += len(want)
        if not ok:
            bad += 1
            os.makedirs("gpurun_out", exist_ok=True)
            open(f"gpurun_out/fuzz_fail_{rounds}.bin", "wb").write(s)
            print(f"FAIL round {rounds}: {[(mt, len(d)) for mt, d in parts]} -> {got[:80]!r}", flush=True)
        if rounds % 10 == 0:
            print(f"{rounds} streams, {nbytes / 1e6:.1f} MB, {bad} bad, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {rounds} streams, {nbytes / 1e6:.1f} MB, {bad} bad", flush=True)
    sys.exit(1 if bad else 0)
