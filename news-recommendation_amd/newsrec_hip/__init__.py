"""MI355X (gfx950) HIP kernels for the two-tower news-recommendation hot path.

`ops` holds the torch.autograd wrappers; `_lib` the ctypes binding of the C ABI
declared in include/newsrec_hip.h.
"""
from ._lib import NewsrecError, EXPORTS, LIB_PATH  # noqa: F401
