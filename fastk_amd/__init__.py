"""fastk_amd -- MI355X (gfx950) k-mer counting engine behind FastK's stage interface.

The product is the C-ABI shared library fastk_amd/lib/libfastk_amd.so (hand-written HIP,
include/fastk_amd.h); this package is its ctypes host mirror.
"""
from .api import (Context, DeviceBuffer, FastKError, Result, HIST_BINS, EXPORTS, LIB_PATH,  # noqa: F401
                  load_library, widths, write_files, Shard)
