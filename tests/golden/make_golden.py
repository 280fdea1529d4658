#!/usr/bin/env python3
"""Regenerates tests/golden/*.zpaq + manifest.json.

The reference (mnadareski/ZPAQSharp) cannot be compiled or run and ships no test
vectors, so these fixtures are produced by this repo's two INDEPENDENT stream
writers — the oracle's encoder mirror (oracle/zpaq_oracle.c) and the product's
CPU writer (zpaqsharp_amd/gen/zpaqgen.cpp) — which must agree byte for byte
before a fixture is written.  Each entry records the plaintext SHA-1, sizes, the
per-4096-bit (p,y) trace digests and the final coder state from the oracle, so a
GPU divergence can be localised.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from tests import util  # noqa: E402
from zpaqsharp_amd import models, synth  # noqa: E402

CASES = [  # (name, model, plaintext kind, bytes)
    ("l1_text_64k", "l1", "text", 65536),
    ("min_text_16k", "min", "text", 16384),
    ("mid_text_16k", "mid", "text", 16384),
    ("max_text_8k", "max", "text", 8192),
    ("max_e8e9_x86_8k", "max+e8e9", "x86", 8192),
    ("l1_random_4k", "l1", "random", 4096),
    ("mid_empty", "mid", "text", 0),
]


def plaintext(kind, n):
    if kind == "text":
        return util.text(n, seed=7)
    if kind == "x86":
        return util.x86ish(n, seed=8)
    return synth.plain("R", 99, n).tobytes()


def main():
    manifest = {}
    for name, model, kind, n in CASES:
        data = plaintext(kind, n)
        a = util.block(model, data)                       # oracle encoder
        b = synth.compress_block(model, data)             # product CPU writer
        assert a == b, f"{name}: the two encoders disagree"
        d = oracle.Decompresser(a)
        d.set_trace()
        assert d.find_block() is not None
        assert d.find_filename() is not None
        d.read_comment()
        out, more = d.decompress(-1, cap=n + 16)
        assert not more and out == data
        sha = d.read_segment_end()
        assert sha == hashlib.sha1(data).digest()
        with open(os.path.join(HERE, name + ".zpaq"), "wb") as f:
            f.write(a)
        manifest[name] = {
            "model": model, "plaintext": kind, "plain_len": n, "stream_len": len(a),
            "plain_sha1": hashlib.sha1(data).hexdigest(), "stream_sha1": hashlib.sha1(a).hexdigest(),
            "header_hex": models.get(model).header.hex(), "pcomp_hex": models.get(model).pcomp.hex(),
            "trace_crc32_per_4096_bits": d.trace(), "final_state_low_high_curr_c8_hmap4_h0_h1_h2": list(d.state()),
        }
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", len(manifest), "fixtures")


if __name__ == "__main__":
    main()
