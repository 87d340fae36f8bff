"""strugatzki_amd -- MI355X (gfx950) implementation of Strugatzki's similarity hot path.

All numerics run in hand-written HIP kernels inside lib/libstrugatzki_amd.so
(C ABI: include/strugatzki_amd.h).  Importing the package loads that library
and fails if it is absent; there is no CPU implementation in here.
"""
from ._native import LIB_PATH, NativeError  # noqa: F401
from .engine import Context, Selector, ring_runs, search_multi, segm_select  # noqa: F401
from .search import SearchParams, search, search_batch, search_punch, search_sharded  # noqa: F401
