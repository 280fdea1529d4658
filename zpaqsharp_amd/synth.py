"""Synthetic ZPAQ streams for benchmarks and full-size tests (libzpaqgen.so).

libzpaqgen is the repo's CPU stream *writer* (Encoder/Compressor mirror,
Encoder.cs:26-103, Compressor.cs:27-299) plus the deterministic plaintext
generators of BASELINE.md §2.  It has no decompression entry point.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

from . import models
from .zpaql import Model

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzpaqgen.so")
KINDS = {"T": 0, "X": 1, "R": 2}
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        vp, sz = C.c_void_p, C.c_size_t
        L.zpaqgen_plain.argtypes = [C.c_int, C.c_uint64, vp, sz]
        L.zpaqgen_plain.restype = None
        L.zpaqgen_e8e9.argtypes = [vp, sz]
        L.zpaqgen_e8e9.restype = None
        L.zpaqgen_sha1.argtypes = [vp, sz, vp]
        L.zpaqgen_sha1.restype = None
        L.zpaqgen_compress_block.argtypes = [vp, sz, vp, sz, vp, sz, vp, sz, C.c_char_p, C.c_char_p, C.c_int, vp, sz,
                                             C.POINTER(sz)]
        L.zpaqgen_compress_block.restype = C.c_long
        L.zpaqgen_stream_new.argtypes = [vp, sz, vp, sz, C.c_int, C.c_int, C.c_uint64, C.c_uint32, sz, C.c_int]
        L.zpaqgen_stream_new.restype = vp
        L.zpaqgen_preprocess.argtypes = [C.POINTER(C.c_int), vp, sz, vp, sz, C.POINTER(sz)]
        L.zpaqgen_preprocess.restype = C.c_long
        L.zpaqgen_method_stream_new.argtypes = [vp, sz, vp, sz, C.POINTER(C.c_int), C.c_int, C.c_uint64, C.c_uint32, sz, C.c_int]
        L.zpaqgen_method_stream_new.restype = vp
        L.zpaqgen_stream_error.argtypes = [vp]
        L.zpaqgen_stream_error.restype = C.c_char_p
        L.zpaqgen_stream_size.argtypes = [vp]
        L.zpaqgen_stream_size.restype = sz
        L.zpaqgen_stream_copy.argtypes = [vp, vp, vp]
        L.zpaqgen_stream_copy.restype = None
        L.zpaqgen_stream_free.argtypes = [vp]
        L.zpaqgen_stream_free.restype = None
        _lib = L
    return _lib


def _u8(b) -> np.ndarray:
    return b if isinstance(b, np.ndarray) else np.frombuffer(bytes(b), np.uint8)


def plain(kind: str, block_index: int, n: int) -> np.ndarray:
    """Plaintext of block `block_index` (splitmix64 seeded 0x5A50415153484152 ^ index)."""
    out = np.empty(n, np.uint8)
    load().zpaqgen_plain(KINDS[kind], block_index, out.ctypes.data, n)
    return out


def e8e9(data) -> np.ndarray:
    a = np.array(_u8(data), copy=True)
    load().zpaqgen_e8e9(a.ctypes.data, a.size)
    return a


def lz77_encode(data) -> np.ndarray:
    """Greedy LZ77 coder for models.LZ77_PCOMP: literal runs (token t < 128: t + 1 bytes) and matches of 3..130
    bytes at distance 1..65535 (token 128 + len - 3, distance low byte, high byte)."""
    d = bytes(_u8(data).tobytes())
    n, out, lits, last, i = len(d), bytearray(), bytearray(), {}, 0

    def flush():
        for k in range(0, len(lits), 128):
            run = lits[k:k + 128]
            out.append(len(run) - 1)
            out.extend(run)
        lits.clear()

    while i < n:
        best = 0
        if i + 3 <= n:
            key = d[i:i + 3]
            j = last.get(key, -1)
            if j >= 0 and i - j <= 65535:
                m = 3
                while m < 130 and i + m < n and d[j + m] == d[i + m]:
                    m += 1
                best, dist = m, i - j
            last[key] = i
        if best >= 3:
            flush()
            out.append(128 + best - 3)
            out.append(dist & 255)
            out.append(dist >> 8)
            for k in range(i + 1, min(i + best, n - 2)):
                last[d[k:k + 3]] = k
            i += best
        else:
            lits.append(d[i])
            i += 1
    flush()
    return np.frombuffer(bytes(out), np.uint8)


def compress_block(model, data, filename: bytes = b"", comment: Optional[bytes] = None, sha1: bool = True,
                   tag: bool = True, pre=None) -> bytes:
    """One block / one segment in LibZPAQ.compressBlock framing (LibZPAQ.cs:296-323).  `pre`: the bytes to code when the
    caller has already run the pre-processor the block's PCOMP inverts (size comment and SHA-1 still describe `data`)."""
    m: Model = models.get(model) if isinstance(model, str) else model
    d = _u8(data)
    src = d
    if pre is not None:
        src = _u8(pre)
    elif m.pcomp_cmd.startswith("e8e9"):
        src = e8e9(d)
    elif m.pcomp_cmd.startswith("lz77"):
        src = lz77_encode(d)
    hdr, pc = _u8(m.header), _u8(m.pcomp) if m.pcomp else None
    cap = max(d.size, src.size) + max(d.size, src.size) // 8 + len(m.header) + 2 * len(m.pcomp) + 4096
    need = C.c_size_t(0)
    for _ in range(2):
        out = np.empty(cap, np.uint8)
        n = load().zpaqgen_compress_block(hdr.ctypes.data, hdr.size, pc.ctypes.data if pc is not None else None,
                                          pc.size if pc is not None else 0, src.ctypes.data, src.size,
                                          d.ctypes.data, d.size, filename, comment,
                                          (1 if sha1 else 0) | (2 if tag else 0), out.ctypes.data, cap, C.byref(need))
        if n == -20:
            cap = need.value
            continue
        if n < 0:
            raise RuntimeError(f"zpaqgen_compress_block failed: {n}")
        return out[:n].tobytes()
    raise RuntimeError("zpaqgen_compress_block: capacity")


def stream(model, kind: str = "T", nblocks: int = 1, block_size: int = 1 << 16, first_block: int = 0,
           threads: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """(stream bytes, block offsets[nblocks+1]) for `nblocks` independent blocks."""
    m: Model = models.get(model) if isinstance(model, str) else model
    L = load()
    hdr, pc = _u8(m.header), _u8(m.pcomp) if m.pcomp else None
    threads = threads or min(32, os.cpu_count() or 1)
    h = L.zpaqgen_stream_new(hdr.ctypes.data, hdr.size, pc.ctypes.data if pc is not None else None,
                             pc.size if pc is not None else 0, KINDS[kind], int(m.pcomp_cmd.startswith("e8e9")),
                             first_block, nblocks, block_size, threads)
    try:
        e = L.zpaqgen_stream_error(h)
        if e:
            raise RuntimeError(e.decode())
        out = np.empty(L.zpaqgen_stream_size(h), np.uint8)
        offs = np.zeros(nblocks + 1, np.uint64)
        L.zpaqgen_stream_copy(h, out.ctypes.data, offs.ctypes.data)
        return out, offs
    finally:
        L.zpaqgen_stream_free(h)


def preprocess(args, data) -> bytes:
    """What the pre-processor of a method (LZBuffer.cs:96-115 formats: level = args[1] & 3) makes of `data`: the fast C++
    twin of tools/methods.preprocess for benchmark-sized inputs (same formats, its own greedy parse)."""
    L = load()
    d = _u8(data)
    a = (C.c_int * 9)(*[int(x) for x in list(args)[:9]] + [0] * (9 - min(9, len(args))))
    need = C.c_size_t(0)
    out = np.empty(max(16, d.size + d.size // 8 + 64), np.uint8)
    rc = L.zpaqgen_preprocess(a, d.ctypes.data if d.size else None, d.size, out.ctypes.data, out.size, C.byref(need))
    if rc == -20:
        out = np.empty(need.value, np.uint8)
        rc = L.zpaqgen_preprocess(a, d.ctypes.data if d.size else None, d.size, out.ctypes.data, out.size, C.byref(need))
    if rc < 0:
        raise RuntimeError(f"zpaqgen_preprocess failed: {rc}")
    return out[:rc].tobytes()


def method_stream(model, args, kind: str = "T", nblocks: int = 1, block_size: int = 1 << 16, first_block: int = 0,
                  threads: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """`nblocks` distinct blocks written with a METHOD of the reference (LibZPAQ.compressBlock framing): plaintext
    generator `kind`, the method's pre-processor (args as tools/methods.make_config returns them), then the model of
    `model` — stored chunks for n = 0.  Returns (stream, block offsets)."""
    L = load()
    hdr, pc = _u8(model.header), _u8(model.pcomp or b"")
    a = (C.c_int * 9)(*[int(x) for x in list(args)[:9]] + [0] * (9 - min(9, len(args))))
    if (int(a[1]) & 3) == 3 and model.pcomp and block_size + 5 > (1 << model.header[5]):
        # (compressBlock never writes such a block; bwtrle's walk over a wrapped M need not end before its instruction budget)
        raise ValueError(f"a BWT block of {block_size} bytes does not fit the post-processor's M (2^{model.header[5]} bytes)")
    if threads is None:
        threads = min(32, os.cpu_count() or 1)
    h = L.zpaqgen_method_stream_new(hdr.ctypes.data, hdr.size, pc.ctypes.data if pc.size else None, pc.size, a, KINDS[kind],
                                    first_block, nblocks, block_size, threads)
    try:
        err = L.zpaqgen_stream_error(h)
        if err:
            raise RuntimeError(err.decode())
        out = np.empty(L.zpaqgen_stream_size(h), np.uint8)
        offs = np.empty(nblocks + 1, np.uint64)
        L.zpaqgen_stream_copy(h, out.ctypes.data, offs.ctypes.data)
        return out, offs
    finally:
        L.zpaqgen_stream_free(h)
