"""ctypes binding of the CPU oracle (oracle/zpaq_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
the cpu_baseline leg of bench.py.  The product package (zpaqsharp_amd/) must
never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzpaq_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "zpaq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libzpaq_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        u8p, sz, vp = C.POINTER(C.c_uint8), C.c_size_t, C.c_void_p
        L.zo_table_pins.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.zo_tables.argtypes = [vp] * 5
        L.zo_dec_new.restype = vp
        L.zo_dec_free.argtypes = [vp]
        L.zo_dec_error.argtypes = [vp]
        L.zo_dec_error.restype = C.c_char_p
        L.zo_dec_set_input.argtypes = [vp, vp, sz]
        L.zo_dec_find_block.argtypes = [vp, C.POINTER(C.c_double)]
        L.zo_dec_find_filename.argtypes = [vp, C.c_char_p, sz]
        L.zo_dec_read_comment.argtypes = [vp, C.c_char_p, sz]
        L.zo_dec_decompress.argtypes = [vp, C.c_long, vp, sz, C.POINTER(sz)]
        L.zo_dec_read_segment_end.argtypes = [vp, vp]
        L.zo_dec_tell.argtypes = [vp]
        L.zo_dec_tell.restype = sz
        L.zo_dec_hcomp.argtypes = [vp, vp, sz]
        L.zo_dec_hcomp.restype = C.c_long
        L.zo_dec_pcomp.argtypes = [vp, vp, sz]
        L.zo_dec_pcomp.restype = C.c_long
        L.zo_dec_set_trace.argtypes = [vp, vp, sz, C.POINTER(sz)]
        L.zo_dec_state.argtypes = [vp, vp]
        L.zo_decompress.argtypes = [vp, sz, vp, sz, C.c_char_p, sz]
        L.zo_decompress.restype = C.c_long
        L.zo_enc_new.argtypes = [vp, sz]
        L.zo_enc_new.restype = vp
        L.zo_enc_free.argtypes = [vp]
        L.zo_enc_error.argtypes = [vp]
        L.zo_enc_error.restype = C.c_char_p
        L.zo_enc_tell.argtypes = [vp]
        L.zo_enc_tell.restype = sz
        L.zo_enc_write_tag.argtypes = [vp]
        L.zo_enc_start_block.argtypes = [vp, vp, sz]
        L.zo_enc_start_segment.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.zo_enc_post_process.argtypes = [vp, vp, sz]
        L.zo_enc_begin_raw.argtypes = [vp]
        L.zo_enc_compress.argtypes = [vp, vp, sz]
        L.zo_enc_end_segment.argtypes = [vp, vp]
        L.zo_enc_end_block.argtypes = [vp]
        L.zo_sha1.argtypes = [vp, sz, vp]
        L.zo_e8e9.argtypes = [vp, sz]
        L.zo_run_pcomp.argtypes = [vp, sz, C.c_int, C.c_int, vp, sz, vp, sz]
        L.zo_run_pcomp.restype = C.c_long
        L.zo_tag_hash.argtypes = [vp, sz, vp]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _buf(b: bytes):
    return (C.c_uint8 * max(1, len(b))).from_buffer_copy(b if b else b"\0")


def table_pins() -> Tuple[int, int, int]:
    a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    lib().zo_table_pins(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def tables():
    import numpy as np
    sq = np.zeros(4096, np.uint16)
    st = np.zeros(32768, np.int16)
    dt = np.zeros(1024, np.int32)
    dt2k = np.zeros(256, np.int32)
    ns = np.zeros(1024, np.uint8)
    lib().zo_tables(sq.ctypes.data, st.ctypes.data, dt.ctypes.data, dt2k.ctypes.data, ns.ctypes.data)
    return sq, st, dt, dt2k, ns


def sha1(data: bytes) -> bytes:
    out = (C.c_uint8 * 20)()
    b = _buf(data)
    lib().zo_sha1(b, len(data), out)
    return bytes(out)


def e8e9(data: bytes) -> bytes:
    b = _buf(data)
    lib().zo_e8e9(b, len(data))
    return bytes(b[:len(data)])


def run_pcomp(pcomp: bytes, data: bytes, ph: int = 0, pm: int = 0, cap: Optional[int] = None) -> bytes:
    cap = cap if cap is not None else len(data) * 2 + 4096
    out = (C.c_uint8 * cap)()
    n = lib().zo_run_pcomp(_buf(pcomp), len(pcomp), ph, pm, _buf(data), len(data), out, cap)
    if n < 0:
        raise OracleError("ZPAQL execution error")
    return bytes(out[:n])


def tag_hash(data: bytes) -> Tuple[int, int, int, int]:
    h = (C.c_uint32 * 4)()
    lib().zo_tag_hash(_buf(data), len(data), h)
    return tuple(h)


def set_zpaql_budget(per_run: int) -> None:
    """Test guard: instructions one ZPAQL run() may execute (0 = unlimited, the reference's behaviour)."""
    lib().zo_set_zpaql_budget(C.c_uint64(per_run))


def decompress(stream: bytes, cap: Optional[int] = None) -> bytes:
    """LibZPAQ.decompress(Reader, Writer) on the oracle."""
    import numpy as np
    cap = cap if cap is not None else max(1 << 16, len(stream) * 64)
    src = np.frombuffer(stream, np.uint8) if len(stream) else np.zeros(1, np.uint8)
    out = np.empty(cap, np.uint8)
    err = C.create_string_buffer(128)
    n = lib().zo_decompress(src.ctypes.data, len(stream), out.ctypes.data, cap, err, 128)
    if n < 0:
        raise OracleError(err.value.decode())
    return out[:n].tobytes()


class Decompresser:
    """Step-wise mirror of the reference's Decompresser call protocol."""

    def __init__(self, stream: bytes):
        self._L = lib()
        self._d = self._L.zo_dec_new()
        self._src = _buf(stream)
        self._L.zo_dec_set_input(self._d, self._src, len(stream))
        self._trace = None

    def close(self):
        if self._d:
            self._L.zo_dec_free(self._d)
            self._d = None

    def __del__(self):
        self.close()

    def _chk(self, r):
        if r < 0:
            raise OracleError(self._L.zo_dec_error(self._d).decode())
        return r

    def find_block(self) -> Optional[float]:
        mem = C.c_double()
        return mem.value if self._chk(self._L.zo_dec_find_block(self._d, C.byref(mem))) else None

    def find_filename(self) -> Optional[bytes]:
        buf = C.create_string_buffer(4096)
        return buf.value if self._chk(self._L.zo_dec_find_filename(self._d, buf, 4096)) else None

    def read_comment(self) -> bytes:
        buf = C.create_string_buffer(4096)
        self._chk(self._L.zo_dec_read_comment(self._d, buf, 4096))
        return buf.value

    def decompress(self, n: int = -1, cap: int = 1 << 24) -> Tuple[bytes, bool]:
        out = (C.c_uint8 * cap)()
        ln = C.c_size_t(0)
        more = self._chk(self._L.zo_dec_decompress(self._d, n, out, cap, C.byref(ln)))
        return bytes(out[:ln.value]), bool(more)

    def read_segment_end(self) -> Optional[bytes]:
        s = (C.c_uint8 * 21)()
        self._chk(self._L.zo_dec_read_segment_end(self._d, s))
        return bytes(s[1:21]) if s[0] else None

    def tell(self) -> int:
        return self._L.zo_dec_tell(self._d)

    def hcomp(self) -> bytes:
        b = (C.c_uint8 * 70000)()
        return bytes(b[:self._L.zo_dec_hcomp(self._d, b, 70000)])

    def pcomp(self) -> bytes:
        b = (C.c_uint8 * 70000)()
        return bytes(b[:self._L.zo_dec_pcomp(self._d, b, 70000)])

    def set_trace(self, cap: int = 1 << 16):
        self._trace = ((C.c_uint32 * cap)(), C.c_size_t(0), cap)
        self._L.zo_dec_set_trace(self._d, self._trace[0], cap, C.byref(self._trace[1]))

    def trace(self) -> List[int]:
        arr, n, cap = self._trace
        return list(arr[:min(n.value, cap)])

    def state(self) -> Tuple[int, ...]:
        st = (C.c_uint32 * 8)()
        self._L.zo_dec_state(self._d, st)
        return tuple(st)


class Compressor:
    """Step-wise mirror of the reference's Compressor (fixture generation)."""

    def __init__(self, cap: int):
        self._L = lib()
        self._out = (C.c_uint8 * cap)()
        self._e = self._L.zo_enc_new(self._out, cap)
        self._cap = cap

    def close(self):
        if self._e:
            self._L.zo_enc_free(self._e)
            self._e = None

    def __del__(self):
        self.close()

    def _chk(self, r):
        if r < 0:
            raise OracleError(self._L.zo_enc_error(self._e).decode() or "oracle encoder overflow")

    def write_tag(self):
        self._chk(self._L.zo_enc_write_tag(self._e))

    def start_block(self, header: bytes):
        self._chk(self._L.zo_enc_start_block(self._e, _buf(header), len(header)))

    def start_segment(self, filename: bytes = b"", comment: bytes = b""):
        self._chk(self._L.zo_enc_start_segment(self._e, filename, comment))

    def post_process(self, pcomp: bytes = b""):
        self._chk(self._L.zo_enc_post_process(self._e, _buf(pcomp) if pcomp else None, len(pcomp)))

    def begin_raw(self):
        """Open the coder without a post-processor selector: the next compress() bytes are what PostProcessor.write sees."""
        self._chk(self._L.zo_enc_begin_raw(self._e))

    def compress(self, data: bytes):
        self._chk(self._L.zo_enc_compress(self._e, _buf(data), len(data)))

    def end_segment(self, sha: Optional[bytes] = None):
        self._chk(self._L.zo_enc_end_segment(self._e, _buf(sha) if sha else None))

    def end_block(self):
        self._chk(self._L.zo_enc_end_block(self._e))

    def getvalue(self) -> bytes:
        n = self._L.zo_enc_tell(self._e)
        if n > self._cap:
            raise OracleError("oracle encoder overflow")
        return bytes(self._out[:n])


def compress_block(header: bytes, data: bytes, pcomp: bytes = b"", filename: bytes = b"",
                   comment: Optional[bytes] = None, with_sha1: bool = True, tag: bool = True) -> bytes:
    """One block, one segment — the framing LibZPAQ.compressBlock produces
    (LibZPAQ.cs:296-323): tag, header, segment(filename, comment=size), sha1."""
    c = Compressor(len(data) * 2 + len(header) + len(pcomp) * 2 + 4096)
    if tag:
        c.write_tag()
    c.start_block(header)
    c.start_segment(filename, str(len(data)).encode() if comment is None else comment)
    # Compressor.postProcess takes pcomp without its trailing END byte count? It
    # sends len = hend-hbegin bytes, which includes the terminating 0 (Compressor.cs:163).
    c.post_process(pcomp)
    c.compress(data)
    c.end_segment(sha1(data) if with_sha1 else None)
    c.end_block()
    out = c.getvalue()
    c.close()
    return out
