"""The C-ABI library loads without a GPU and exports every symbol its header declares."""
import ctypes
import os
import re

from common import ROOT

import strugatzki_amd
from strugatzki_amd import _native as N

HEADER = os.path.join(ROOT, "include", "strugatzki_amd.h")


def _declared():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"SGZ_API[^;(]*?\b(sgz_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    names = _declared()
    assert len(names) >= 30
    lib = ctypes.CDLL(N.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"not exported: {missing}"


def test_python_binding_covers_the_header():
    assert sorted(N.SIGNATURES) == _declared()


def test_entry_points_cite_the_reference():
    """Every declaration is preceded by a comment; those that replace reference code name a .scala file."""
    text = open(HEADER).read()
    assert text.count(".scala:") >= 15


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        return  # covered by the GPU tests
    try:
        strugatzki_amd.Context(0)
    except strugatzki_amd.NativeError as e:
        assert e.code == N.SGZ_E_HIP
    else:
        raise AssertionError("context creation must fail without a GPU")


def test_version_string():
    assert b"gfx950" in N.lib.sgz_version()


def test_null_and_bad_arguments_are_refused_without_a_gpu():
    """Argument checks of the newer entry points come before any GPU call."""
    import ctypes as C
    lib = N.lib
    n = C.c_int(0)
    assert lib.sgz_corr_search(None, 10, 1, 1, 0, 512, N.SGZ_SCREEN_AUTO, 1, 1024, -1.0, None, 0, C.byref(n)) == N.SGZ_E_ARG
    assert lib.sgz_set_kernel_timing(None, 1) == N.SGZ_E_ARG
    assert lib.sgz_corr_set_tail_trim(None, 0) == N.SGZ_E_ARG
    assert lib.sgz_kernel_ms_history(None, None, 0) == N.SGZ_E_ARG
    assert lib.sgz_sel_feed_gathered(None, None, 1, 1, None, None, 0, 0) == N.SGZ_E_ARG
