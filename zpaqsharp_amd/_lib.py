"""ctypes binding of libzpaqhip.so (include/zpaqhip.h).

The library is built in-tree (zpaqsharp_amd/libzpaqhip.so) by
`__graft_entry__.build()` / `make -C zpaqsharp_amd/csrc`.  Loading fails loudly
if it is missing: there is no Python or CPU fallback for the decode path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzpaqhip.so")

UINT64_MAX = (1 << 64) - 1


class Err(C.Structure):
    _fields_ = [("code", C.c_int32), ("block", C.c_int32), ("segment", C.c_int32), ("msg", C.c_char * 116)]


class Block(C.Structure):
    _fields_ = [("tag_off", C.c_uint64), ("hdr_off", C.c_uint64), ("hdr_len", C.c_uint32),
                ("level", C.c_uint8), ("n_comp", C.c_uint8), ("hh", C.c_uint8), ("hm", C.c_uint8),
                ("ph", C.c_uint8), ("pm", C.c_uint8), ("reserved", C.c_uint16),
                ("first_seg", C.c_uint32), ("n_seg", C.c_uint32), ("end_off", C.c_uint64),
                ("model_mem", C.c_double), ("usize_hint", C.c_uint64)]


class Segment(C.Structure):
    _fields_ = [("block", C.c_uint32), ("flags", C.c_uint32), ("name_off", C.c_uint64),
                ("name_len", C.c_uint32), ("comment_len", C.c_uint32), ("comment_off", C.c_uint64),
                ("data_off", C.c_uint64), ("data_len", C.c_uint64), ("usize_hint", C.c_uint64),
                ("sha1", C.c_uint8 * 20), ("reserved", C.c_uint32)]


class SegResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("pp_state", C.c_uint32), ("out_off", C.c_uint64), ("out_len", C.c_uint64),
                ("in_used", C.c_uint64)]


class Opts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("verify_sha1", C.c_uint32), ("max_concurrent", C.c_uint32),
                ("kernel", C.c_uint32), ("zpaql_budget", C.c_uint64), ("batch_blocks", C.c_uint64),
                ("queue_blocks", C.c_uint64), ("reserved", C.c_uint64 * 2)]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("init_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
                ("blocks", C.c_uint64), ("in_bytes", C.c_uint64), ("out_bytes", C.c_uint64),
                ("model_bytes", C.c_uint64), ("launches", C.c_uint32), ("concurrent", C.c_uint32),
                ("kernel_kind", C.c_uint32), ("reserved", C.c_uint32)]


READ_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_int)
WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_int)

# Every symbol include/zpaqhip.h declares; tests check the library exports them all.
SYMBOLS = ("zpaqhip_version", "zpaqhip_strerror", "zpaqhip_device_count", "zpaqhip_ctx_create",
           "zpaqhip_ctx_destroy", "zpaqhip_last_stats", "zpaqhip_scan", "zpaqhip_decompress",
           "zpaqhip_decompress_segments", "zpaqhip_decompress_cb", "zpaqhip_decode_blocks_device", "zpaqhip_read_device_tables",
           "zpaqhip_block_pcomp", "zpaqhip_decompress_multi", "zpaqhip_decompress_multi_stats", "zpaqhip_block_costs", "zpaqhip_multi_trim")

_lib = None


def load():
    """Load libzpaqhip.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the ZPAQ decode path)")
    # One HIP runtime per process: PyTorch ships its own libamdhip64; importing it first makes
    # libzpaqhip.so bind to that copy, so tensors and this library share devices and streams.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, errp = C.c_void_p, C.c_size_t, C.POINTER(Err)
    L.zpaqhip_version.restype = C.c_int
    L.zpaqhip_strerror.argtypes = [C.c_int]
    L.zpaqhip_strerror.restype = C.c_char_p
    L.zpaqhip_device_count.restype = C.c_int
    L.zpaqhip_ctx_create.argtypes = [C.c_int, C.POINTER(vp), errp]
    L.zpaqhip_ctx_destroy.argtypes = [vp]
    L.zpaqhip_ctx_destroy.restype = None
    L.zpaqhip_last_stats.argtypes = [vp, C.POINTER(Stats)]
    L.zpaqhip_scan.argtypes = [vp, sz, C.POINTER(Block), sz, C.POINTER(sz), C.POINTER(Segment), sz, C.POINTER(sz), errp]
    L.zpaqhip_decompress.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(Opts), errp]
    L.zpaqhip_decompress_segments.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(SegResult), sz, C.POINTER(sz),
                                              C.POINTER(Opts), errp]
    L.zpaqhip_decompress_cb.argtypes = [vp, READ_FN, WRITE_FN, vp, C.POINTER(Opts), errp]
    L.zpaqhip_decode_blocks_device.argtypes = [vp, vp, vp, sz, C.POINTER(Block), sz, C.POINTER(Segment), sz,
                                               C.POINTER(C.c_uint32), sz, vp, C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_uint64), C.POINTER(SegResult), C.POINTER(Opts), vp, errp]
    L.zpaqhip_read_device_tables.argtypes = [vp, vp, vp, vp, vp, vp, errp]
    L.zpaqhip_block_pcomp.argtypes = [vp, vp, sz, C.c_uint32, vp, sz, C.POINTER(sz), errp]
    L.zpaqhip_decompress_multi.argtypes = [C.POINTER(C.c_int), sz, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(Opts), errp]
    L.zpaqhip_decompress_multi_stats.argtypes = [C.POINTER(C.c_int), sz, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(Opts),
                                                 C.POINTER(Stats), errp]
    L.zpaqhip_block_costs.argtypes = [vp, sz, C.POINTER(Block), sz, C.POINTER(Segment), sz, vp, errp]
    L.zpaqhip_multi_trim.restype = None
    _lib = L
    return L
