"""zpaqsharp_amd — MI355X-native ZPAQ block decompression (drop-in for the
ZPAQSharp `Decompresser` hot path).  See DESIGN.md and include/zpaqhip.h."""
from .api import (Context, ScanResult, ZpaqError, block_costs, decompress_multi, device_count, make_opts, multi_trim, scan, strerror,  # noqa: F401
                  version)

__all__ = ["Context", "ScanResult", "ZpaqError", "block_costs", "decompress_multi", "device_count", "make_opts", "multi_trim", "scan", "strerror", "version"]
