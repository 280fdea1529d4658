"""One-off fuzzer for the two-wave single-CM kernel (not collected by pytest): random alphabets, run lengths,
block sizes, shifts, segment structure; GPU output vs the oracle.  Usage: python tests/fuzz_l1.py [cases] [seed] [kernel]
(kernel 6: the two-blocks-per-workgroup form)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
import zpaqsharp_amd as z  # noqa: E402
from zpaqsharp_amd import synth, zpaql  # noqa: E402


def sample(rng, n):
    kind = rng.integers(0, 5)
    if kind == 0:                                   # small alphabet, zipf-ish
        k = int(rng.integers(2, 80))
        p = 1.0 / np.arange(1, k + 1) ** rng.uniform(0.5, 2.0)
        return rng.choice(rng.permutation(256)[:k].astype(np.uint8), size=n, p=p / p.sum()).tobytes()
    if kind == 1:                                   # runs
        r = int(rng.integers(1, 40))
        return np.repeat(rng.integers(0, 256, n // r + 1, dtype=np.uint8), r)[:n].tobytes()
    if kind == 2:                                   # uniform random
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 3:                                   # order-1 structure: next = f(prev) + noise
        out = np.empty(n, np.uint8)
        c = 0
        tab = rng.integers(0, 256, 256)
        noise = rng.random(n) < rng.uniform(0.0, 0.3)
        rnd = rng.integers(0, 256, n)
        for i in range(n):
            c = int(rnd[i]) if noise[i] else int(tab[c])
            out[i] = c
        return out.tobytes()
    return (bytes(rng.integers(97, 123, 64, dtype=np.uint8)) * (n // 64 + 1))[:n]   # periodic


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = z.Context(0)
    kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    for case in range(cases):
        bits, shift = int(rng.integers(9, 23)), int(rng.integers(9, 20))
        m = zpaql.assemble(f"comp 0 0 0 0 1 0 cm {bits} {int(rng.integers(1, 256))} hcomp a<<= {shift} *d=a halt end")
        nseg = int(rng.choice([1, 1, 1, 2, 5]))
        parts = [sample(rng, int(rng.integers(0, 120000))) for _ in range(nseg)]
        c = oracle.Compressor(sum(map(len, parts)) * 2 + 100000)
        c.write_tag(); c.start_block(m.header)
        for i, part in enumerate(parts):
            c.start_segment(b"f%d" % i, str(len(part)).encode())
            if i == 0:
                c.post_process(m.pcomp)
            c.compress(part)
            c.end_segment(oracle.sha1(part))
        c.end_block()
        s = c.getvalue()
        want = b"".join(parts)
        got = ctx.decompress(s, verify_sha1=True, kernel=kernel).tobytes()
        assert got == want, (case, bits, shift, nseg, [len(p) for p in parts])
        print("case", case, "ok", bits, shift, nseg, len(want), flush=True)
    print("all", cases, "cases ok")


if __name__ == "__main__":
    main()
