"""Thin Python layer over the C ABI (include/zpaqhip.h): scan, Context, errors.

Everything that decodes goes through libzpaqhip.so and therefore through the
HIP kernels; nothing here implements or falls back to a CPU decoder.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import Block, Err, Opts, SegResult, Segment, Stats, UINT64_MAX


class ZpaqError(RuntimeError):
    """Raised for any non-zero zpaqhip_status; `.code`, `.block`, `.segment` say where."""

    def __init__(self, code: int, block: int = -1, segment: int = -1, msg: str = ""):
        self.code, self.block, self.segment = code, block, segment
        super().__init__(msg or _lib.load().zpaqhip_strerror(code).decode())


def _raise(err: Err, rc: int):
    raise ZpaqError(rc, err.block, err.segment, err.msg.decode(errors="replace"))


def _as_u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        a = data.view(np.uint8).reshape(-1)
        return a if a.flags.c_contiguous else np.ascontiguousarray(a)
    return np.frombuffer(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data, dtype=np.uint8)


def version() -> int:
    return _lib.load().zpaqhip_version()


def device_count() -> int:
    return _lib.load().zpaqhip_device_count()


def strerror(code: int) -> str:
    return _lib.load().zpaqhip_strerror(code).decode()


@dataclass
class ScanResult:
    blocks: "C.Array[Block]"
    segments: "C.Array[Segment]"

    @property
    def n_blocks(self) -> int:
        return len(self.blocks)

    @property
    def n_segments(self) -> int:
        return len(self.segments)


def scan(stream, partial: bool = False):
    """Host-side framing scan (Decompresser.findBlock/findFilename/readComment/
    readSegmentEnd, Decompresser.cs:29-108,163-194) → block and segment tables.
    partial=True: a framing error does not raise; the blocks parsed before the damage are returned together with
    the error, (ScanResult, ZpaqError | None) — the reference's Decompresser delivers those blocks too."""
    L = _lib.load()
    a = _as_u8(stream)
    nb, ns, err = C.c_size_t(0), C.c_size_t(0), Err()
    rc = L.zpaqhip_scan(a.ctypes.data, a.size, None, 0, C.byref(nb), None, 0, C.byref(ns), C.byref(err))
    if rc and not partial:
        _raise(err, rc)
    blocks = (Block * max(1, nb.value))()
    segs = (Segment * max(1, ns.value))()
    rc = L.zpaqhip_scan(a.ctypes.data, a.size, blocks, nb.value, C.byref(nb), segs, ns.value, C.byref(ns), C.byref(err))
    if rc and not partial:
        _raise(err, rc)
    res = ScanResult((Block * nb.value).from_buffer(blocks) if nb.value else (Block * 0)(),
                     (Segment * ns.value).from_buffer(segs) if ns.value else (Segment * 0)())
    if partial:
        return res, (ZpaqError(rc, err.block, err.segment, err.msg.decode(errors="replace")) if rc else None)
    return res


def block_costs(stream, sc: "ScanResult") -> np.ndarray:
    """zpaqhip_block_costs: estimated decode cost per block (plaintext bytes x cycles per byte of the block's
    kernel), the weight of every multi-GPU plan (multigpu.py, zpaqhip_decompress_multi).  Host-side."""
    L = _lib.load()
    a = _as_u8(stream)
    cost = np.zeros(max(1, sc.n_blocks), np.uint64)
    err = Err()
    rc = L.zpaqhip_block_costs(a.ctypes.data, a.size, sc.blocks, sc.n_blocks, sc.segments, sc.n_segments, cost.ctypes.data, C.byref(err))
    if rc:
        _raise(err, rc)
    return cost[:sc.n_blocks]


def decompress_multi(devices: Sequence[int], stream, out_cap: Optional[int] = None, per_device: Optional[list] = None,
                     partial: bool = False, **opt):
    """zpaqhip_decompress_multi(_stats): LibZPAQ.decompress over several GPUs of one node (one context + host thread
    per entry of `devices`; an entry may repeat), the threads pulling chunks of `queue_blocks` blocks from one
    cost-ordered work queue; plaintext in stream order.  per_device: a list that receives one Stats per device.
    partial=True: a damaged block does not raise; returns (plaintext before it, ZpaqError | None)."""
    L = _lib.load()
    a = _as_u8(stream)
    o = make_opts(**opt)
    err, n = Err(), C.c_size_t(0)
    devs = (C.c_int * len(devices))(*devices)
    if out_cap is None:
        sc = scan(a, partial=True)[0]
        out_cap = 0
        for b in sc.blocks:
            coded = sum(int(sc.segments[b.first_seg + i].data_len) for i in range(b.n_seg))
            out_cap += int(b.usize_hint) if b.usize_hint != UINT64_MAX and b.usize_hint <= 1 << 40 else 8 * coded + (64 << 10)
    st = (Stats * len(devices))()
    for _ in range(2):
        out = np.empty(max(1, out_cap), np.uint8)
        rc = L.zpaqhip_decompress_multi_stats(devs, len(devices), a.ctypes.data, a.size, out.ctypes.data, out_cap, C.byref(n),
                                              C.byref(o), st, C.byref(err))
        if not (rc == -20 and n.value > out_cap):
            break
        out_cap = n.value
    if per_device is not None:
        per_device[:] = [Stats.from_buffer_copy(bytes(x)) for x in st]
    if partial:
        e = ZpaqError(rc, err.block, err.segment, err.msg.decode(errors="replace")) if rc else None
        return out[:min(n.value, out_cap)], e
    if rc:
        _raise(err, rc)
    return out[:n.value]


def multi_trim() -> None:
    """zpaqhip_multi_trim: destroy the contexts decompress_multi keeps between calls."""
    _lib.load().zpaqhip_multi_trim()


def make_opts(verify_sha1: bool = False, max_concurrent: int = 0, kernel: int = 0, zpaql_budget: int = 0,
              batch_blocks: int = 0, queue_blocks: int = 0) -> Opts:
    o = Opts()
    o.struct_size = C.sizeof(Opts)
    o.verify_sha1 = int(verify_sha1)
    o.max_concurrent = max_concurrent
    o.kernel = kernel
    o.zpaql_budget = zpaql_budget
    o.batch_blocks = batch_blocks
    o.queue_blocks = queue_blocks
    return o


class Context:
    """One zpaqhip_ctx: bound to one GPU, not thread-safe (LICENSE:44-46 contract)."""

    def __init__(self, device: int = 0):
        self._L = _lib.load()
        h, err = C.c_void_p(), Err()
        rc = self._L.zpaqhip_ctx_create(device, C.byref(h), C.byref(err))
        if rc:
            _raise(err, rc)
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.zpaqhip_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        self.close()

    def stats(self) -> Stats:
        s = Stats()
        self._L.zpaqhip_last_stats(self._h, C.byref(s))
        return s

    def device_tables(self):
        """(squash, stretch, dt, dt2k, ns) as read back from device memory."""
        sq, st = np.zeros(4096, np.uint16), np.zeros(32768, np.int16)
        dt, dt2k, ns = np.zeros(1024, np.int32), np.zeros(256, np.int32), np.zeros(1024, np.uint8)
        err = Err()
        rc = self._L.zpaqhip_read_device_tables(self._h, sq.ctypes.data, st.ctypes.data, dt.ctypes.data,
                                                dt2k.ctypes.data, ns.ctypes.data, C.byref(err))
        if rc:
            _raise(err, rc)
        return sq, st, dt, dt2k, ns

    def decompress(self, stream, out_cap: Optional[int] = None, **opt) -> np.ndarray:
        """LibZPAQ.decompress(Reader, Writer) (LibZPAQ.cs:65-79) on host buffers."""
        a = _as_u8(stream)
        o = make_opts(**opt)
        err, n = Err(), C.c_size_t(0)
        if out_cap is None:
            sc = scan(a)
            hints = [b.usize_hint for b in sc.blocks]
            out_cap = sum(h for h in hints if h != UINT64_MAX) if hints and all(h != UINT64_MAX for h in hints) else 0
        out = np.empty(max(1, out_cap), np.uint8)
        rc = self._L.zpaqhip_decompress(self._h, a.ctypes.data, a.size, out.ctypes.data, out_cap, C.byref(n), C.byref(o), C.byref(err))
        if rc == -20 and n.value > out_cap:             # ZPAQHIP_E_OUTPUT_FULL: now we know the size
            out_cap = n.value
            out = np.empty(max(1, out_cap), np.uint8)
            rc = self._L.zpaqhip_decompress(self._h, a.ctypes.data, a.size, out.ctypes.data, out_cap, C.byref(n), C.byref(o), C.byref(err))
        if rc:
            _raise(err, rc)
        return out[:n.value]

    def decompress_into(self, stream: np.ndarray, out: np.ndarray, **opt) -> int:
        """zpaqhip_decompress on caller-owned host buffers (e.g. pinned memory): returns the plaintext length."""
        o = make_opts(**opt)
        err, n = Err(), C.c_size_t(0)
        rc = self._L.zpaqhip_decompress(self._h, stream.ctypes.data, stream.size, out.ctypes.data, out.size, C.byref(n), C.byref(o), C.byref(err))
        if rc:
            _raise(err, rc)
        return n.value

    def block_pcomp(self, stream, block: int) -> bytes:
        """Decompresser.pcomp() (Decompresser.cs:155-158): b"" if block `block` has no PCOMP, else
        length-lo, length-hi, program bytes (ZPAQL.write(out, true), ZPAQL.cs:171-177)."""
        a = _as_u8(stream)
        err, n = Err(), C.c_size_t(0)
        out = np.empty(65536 + 2, np.uint8)
        rc = self._L.zpaqhip_block_pcomp(self._h, a.ctypes.data, a.size, block, out.ctypes.data, out.size, C.byref(n), C.byref(err))
        if rc:
            _raise(err, rc)
        return out[:n.value].tobytes()

    def decompress_segments(self, stream, **opt):
        """Whole stream, but per-segment outcomes are returned instead of raised:
        (plaintext ndarray, SegResult array indexed like scan(stream).segments)."""
        a = _as_u8(stream)
        o = make_opts(**opt)
        err, n, nr = Err(), C.c_size_t(0), C.c_size_t(0)
        res = (SegResult * 1)()
        out = np.empty(1, np.uint8)
        cap, rcap = 0, 0
        for _ in range(3):
            rc = self._L.zpaqhip_decompress_segments(self._h, a.ctypes.data, a.size, out.ctypes.data, cap, C.byref(n),
                                                     res, rcap, C.byref(nr), C.byref(o), C.byref(err))
            if rc in (-20, -25) and (n.value > cap or nr.value > rcap):   # learn sizes, then retry
                cap, rcap = max(cap, n.value), max(rcap, nr.value)
                out = np.empty(max(1, cap), np.uint8)
                res = (SegResult * max(1, rcap))()
                continue
            break
        # a framing error behind the last good block: the results of the blocks before it are valid; the caller finds
        # the error in `framing_error` and raises it on reaching that point (as the reference would)
        self.framing_error = None
        if rc in (-2, -5, -11, -12, -13, -14, -15) and nr.value <= rcap and n.value <= cap:
            self.framing_error = ZpaqError(rc, err.block, err.segment, err.msg.decode(errors="replace"))
            rc = 0
        if rc:
            _raise(err, rc)
        return out[:n.value], res

    def decompress_cb(self, read_fn, write_fn, **opt) -> None:
        """Streaming form: read_fn(n)->bytes ('' at EOF), write_fn(bytes)."""
        o = make_opts(**opt)
        err = Err()
        exc: List[BaseException] = []

        def _r(_u, buf, n):
            try:
                b = read_fn(n)
                C.memmove(buf, b, len(b))
                return len(b)
            except BaseException as e:                  # noqa: BLE001 - must not unwind through C
                exc.append(e)
                return -1

        def _w(_u, buf, n):
            try:
                write_fn(C.string_at(buf, n))
                return 0
            except BaseException as e:                  # noqa: BLE001
                exc.append(e)
                return -1

        rc = self._L.zpaqhip_decompress_cb(self._h, _lib.READ_FN(_r), _lib.WRITE_FN(_w), None, C.byref(o), C.byref(err))
        if exc:
            raise exc[0]
        if rc:
            _raise(err, rc)

    def decode_blocks_device(self, d_in: int, in_len: int, sc: ScanResult, d_out: int,
                             out_off: Sequence[int], out_cap: Sequence[int], ids: Optional[Sequence[int]] = None,
                             h_in=None, hip_stream: int = 0, raise_on_error: bool = True, **opt):
        """Explicit block-table form on device buffers (pointers as ints)."""
        o = make_opts(**opt)
        err = Err()
        nsel = len(ids) if ids is not None else sc.n_blocks
        ids_a = (C.c_uint32 * max(1, nsel))(*ids) if ids is not None else None
        off_a = (C.c_uint64 * max(1, nsel))(*out_off)
        cap_a = (C.c_uint64 * max(1, nsel))(*out_cap)
        res = (SegResult * max(1, sc.n_segments))()
        h = _as_u8(h_in) if h_in is not None else None
        rc = self._L.zpaqhip_decode_blocks_device(
            self._h, d_in, h.ctypes.data if h is not None else None, in_len, sc.blocks, sc.n_blocks, sc.segments,
            sc.n_segments, ids_a, nsel if ids is not None else 0, d_out, off_a, cap_a, res, C.byref(o),
            hip_stream or None, C.byref(err))
        if rc and raise_on_error:
            _raise(err, rc)
        return rc, res
