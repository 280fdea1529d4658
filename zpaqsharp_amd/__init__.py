"""zpaqsharp_amd — MI355X-native ZPAQ block decompression (drop-in for the
ZPAQSharp `Decompresser` hot path).  See DESIGN.md and include/zpaqhip.h."""
from .api import Context, ScanResult, ZpaqError, device_count, make_opts, scan, strerror, version  # noqa: F401

__all__ = ["Context", "ScanResult", "ZpaqError", "device_count", "make_opts", "scan", "strerror", "version"]
