"""Shared helpers for the tests: deterministic plaintext and stream builders.
Streams are produced by the oracle's encoder mirror (test infrastructure)."""
import numpy as np

import oracle
from zpaqsharp_amd import models


def text(n: int, seed: int = 1) -> bytes:
    rng = np.random.default_rng(seed)
    vocab = [bytes(rng.integers(97, 123, rng.integers(2, 10)).astype(np.uint8)) for _ in range(512)]
    p = 1.0 / np.arange(1, 513) ** 1.1
    p /= p.sum()
    out = bytearray()
    while len(out) < n:
        ids = rng.choice(512, 4096, p=p)
        k = 0
        for i in ids:
            out += vocab[i]
            k += 1
            out += b". \n" if k % 12 == 0 else b" "
    return bytes(out[:n])


def x86ish(n: int, seed: int = 2) -> bytes:
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, n, dtype=np.uint8)
    a[rng.random(n) < 0.5] = 0x8B
    for i in range(0, n - 8, 23):
        a[i] = 0xE8 if (i // 23) % 2 else 0xE9
        a[i + 1:i + 4] = rng.integers(0, 256, 3, dtype=np.uint8)
        a[i + 4] = 0 if (i // 23) % 3 else 0xFF
    return a.tobytes()


def block(model: str, data: bytes, **kw) -> bytes:
    """One block in LibZPAQ.compressBlock framing, made by the oracle's encoder."""
    m = models.get(model)
    if m.pcomp:
        enc_in = oracle.e8e9(data) if "e8e9" in model else data
        c = oracle.Compressor(len(data) * 2 + 8192)
        c.write_tag()
        c.start_block(m.header)
        c.start_segment(kw.get("filename", b""), str(len(data)).encode())
        c.post_process(m.pcomp)
        c.compress(enc_in)
        c.end_segment(oracle.sha1(data))
        c.end_block()
        out = c.getvalue()
        c.close()
        return out
    return oracle.compress_block(m.header, data, **kw)
