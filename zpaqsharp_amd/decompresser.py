"""Host-side mirror of the reference's operator interface for this path:
`Reader`, `Writer`, `Decompresser` and `decompress(Reader, Writer)` with the same
names, call order, argument meaning and error behaviour as
Reader.cs:7-28, Writer.cs:12-29, Decompresser.cs:11-221 and LibZPAQ.cs:65-79.

The reference decodes byte by byte on the CPU as the caller pulls; here the whole
input is read from the Reader once, every block is decoded on the GPU in one
launch (read-ahead), and the documented call sequence

    d.setInput(r); while d.findBlock(): while d.findFilename(w): d.readComment(w);
        d.setOutput(out); d.decompress(); d.readSegmentEnd()

is then served from that result — including `decompress(n)` resumption and the
errors, which are raised when the caller reaches the failing segment, as the
reference would.  No CPU decoder is involved.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from . import api
from .api import ZpaqError


class Reader:
    """Reader.cs:7-28 — derive and override get() (and optionally read())."""

    def get(self) -> int:                    # 0..255, or -1 at EOF
        return -1

    def read(self, n: int) -> bytes:         # default: n calls of get()
        out = bytearray()
        while len(out) < n:
            c = self.get()
            if c < 0:
                break
            out.append(c)
        return bytes(out)


class Writer:
    """Writer.cs:12-29 — derive and override put() (and optionally write())."""

    def put(self, c: int) -> None:
        pass

    def write(self, buf: bytes) -> None:     # default: one put() per byte
        for c in buf:
            self.put(c)


class BytesReader(Reader):
    def __init__(self, data: bytes):
        self._d, self._p = memoryview(data), 0

    def get(self) -> int:
        if self._p >= len(self._d):
            return -1
        self._p += 1
        return self._d[self._p - 1]

    def read(self, n: int) -> bytes:
        b = bytes(self._d[self._p:self._p + n])
        self._p += len(b)
        return b


class BytesWriter(Writer):
    def __init__(self):
        self.buf = bytearray()

    def put(self, c: int) -> None:
        self.buf.append(c)

    def write(self, buf: bytes) -> None:
        self.buf += buf


_BLOCK, _FILENAME, _COMMENT, _DATA, _SEGEND = range(5)     # Decompresser.cs:212-215


class Decompresser:
    def __init__(self, context: Optional[api.Context] = None, device: int = 0):
        self._ctx = context or api.Context(device)
        self._own = context is None
        self._in: Optional[Reader] = None
        self._out: Optional[Writer] = None
        self._sha1 = None
        self._state = _BLOCK
        self._stream: Optional[np.ndarray] = None
        self._scan: Optional[api.ScanResult] = None
        self._scan_err: Optional[ZpaqError] = None
        self._res = None
        self._plain: Optional[np.ndarray] = None
        self._b = -1                          # current block
        self._s = -1                          # current segment (global index)
        self._pos = 0                         # bytes of the current segment already delivered

    # ---- Decompresser.cs:22-25
    def setInput(self, reader: Reader) -> None:
        self._in = reader
        self._stream = None

    def _load(self):
        if self._stream is not None:
            return
        chunks: List[bytes] = []
        while True:                           # Decoder.get -> Reader.read (Decoder.cs:112-122)
            b = self._in.read(1 << 16) if self._in else b""
            if not b:
                break
            chunks.append(b)
        self._stream = np.frombuffer(b"".join(chunks), np.uint8)
        # framing damage: the blocks before it are kept, the error is raised when the caller gets there
        self._scan, self._scan_err = api.scan(self._stream, partial=True)
        self._b = -1

    def _decode_all(self):
        if self._res is None:
            self._plain, self._res = self._ctx.decompress_segments(self._stream)

    # ---- Decompresser.cs:29-58
    def findBlock(self) -> bool:
        assert self._state == _BLOCK
        self._load()
        if self._b + 1 >= self._scan.n_blocks:
            if self._scan_err is not None:
                e, self._scan_err = self._scan_err, None
                raise e
            return False
        self._b += 1
        blk = self._scan.blocks[self._b]
        self._s = blk.first_seg - 1
        self._seg_end = blk.first_seg + blk.n_seg
        self._state = _FILENAME
        return True

    def memory(self) -> float:               # *memptr of findBlock (ZPAQL.memory, ZPAQL.cs:58-81)
        return self._scan.blocks[self._b].model_mem

    def hcomp(self, out: Writer) -> None:    # Decompresser.cs:60-63
        b = self._scan.blocks[self._b]
        out.write(self._stream[b.hdr_off:b.hdr_off + b.hdr_len].tobytes())

    # ---- Decompresser.cs:67-93
    def findFilename(self, filename: Optional[Writer] = None) -> bool:
        assert self._state == _FILENAME
        if self._s + 1 >= self._seg_end:
            self._state = _BLOCK
            return False
        self._s += 1
        g = self._scan.segments[self._s]
        if filename is not None:
            filename.write(self._stream[g.name_off:g.name_off + g.name_len].tobytes())
        self._state = _COMMENT
        return True

    # ---- Decompresser.cs:96-108
    def readComment(self, comment: Optional[Writer] = None) -> None:
        assert self._state == _COMMENT
        g = self._scan.segments[self._s]
        if comment is not None:
            comment.write(self._stream[g.comment_off:g.comment_off + g.comment_len].tobytes())
        self._pos = 0
        self._state = _DATA

    def setOutput(self, out: Optional[Writer]) -> None:      # Decompresser.cs:110-113
        self._out = out

    def setSHA1(self, sha1) -> None:                          # Decompresser.cs:115-118 (hashlib-style .update)
        self._sha1 = sha1

    # ---- Decompresser.cs:121-153
    def decompress(self, n: int = -1) -> bool:
        assert self._state == _DATA
        self._decode_all()
        r = self._res[self._s]
        if r.status != 0:
            raise ZpaqError(r.status, self._b, self._s)
        left = r.out_len - self._pos
        take = left if n < 0 else min(n, left)
        if take:
            chunk = self._plain[r.out_off + self._pos:r.out_off + self._pos + take].tobytes()
            if self._out is not None:
                self._out.write(chunk)
            if self._sha1 is not None:
                self._sha1.update(chunk)
            self._pos += take
        if n < 0 or take < n:                  # the decoder met EOS inside this call
            self._state = _SEGEND
            return False
        return True

    def pcomp(self, out: Writer) -> bool:     # Decompresser.cs:155-158: false when the block has no PCOMP
        prog = self._ctx.block_pcomp(self._stream, self._b)
        if prog:
            out.write(prog)
        return bool(prog)

    # ---- Decompresser.cs:163-194
    def readSegmentEnd(self) -> Optional[bytes]:
        """Returns the 20-byte stored SHA-1, or None (sha1string[0] == 0)."""
        assert self._state in (_DATA, _SEGEND)
        g = self._scan.segments[self._s]
        self._state = _FILENAME
        return bytes(g.sha1) if g.flags & 1 else None

    def stat(self, x: int) -> int:            # Decompresser.cs:196-199 (stub in the reference)
        return 0

    def buffered(self) -> int:                # Decompresser.cs:201-204: bytes read ahead of the current position
        if self._stream is None or self._s < 0:
            return 0
        g = self._scan.segments[self._s]
        return int(self._stream.size - (g.data_off + g.data_len))

    def close(self):
        if self._own:
            self._ctx.close()


def decompress(reader: Reader, writer: Writer, context: Optional[api.Context] = None) -> None:
    """LibZPAQ.decompress(Reader in, Writer out), LibZPAQ.cs:65-79."""
    d = Decompresser(context)
    try:
        d.setInput(reader)
        d.setOutput(writer)
        while d.findBlock():
            while d.findFilename():
                d.readComment()
                d.decompress()
                d.readSegmentEnd()
    finally:
        d.close()
