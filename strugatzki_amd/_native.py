"""ctypes binding of libstrugatzki_amd.so (the C ABI declared in include/strugatzki_amd.h).

The library is the product: if it is missing or does not load, importing this
module raises -- there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGZ_LIB_PATH") or os.path.join(_HERE, "lib", "libstrugatzki_amd.so")

SGZ_OK = 0
SGZ_E_ARG = -1
SGZ_E_HIP = -2
SGZ_E_STATE = -3
SGZ_E_OVERFLOW = -4
SGZ_E_IO = -5
SGZ_MAX_CHANNELS = 64
SGZ_SCREEN_NONE, SGZ_SCREEN_F32, SGZ_SCREEN_F16, SGZ_SCREEN_AUTO = 0, 1, 2, 3


class Candidate(C.Structure):
    _fields_ = [("file", C.c_int32), ("offset", C.c_int32), ("sim", C.c_float), ("boost", C.c_float)]


class Match(C.Structure):
    _fields_ = [
        ("sim", C.c_float),
        ("boost_in", C.c_float),
        ("boost_out", C.c_float),
        ("file", C.c_int32),
        ("start", C.c_int64),
        ("stop", C.c_int64),
    ]


class Break(C.Structure):
    _fields_ = [("sim", C.c_float), ("pad_", C.c_int32), ("pos", C.c_int64)]


class Plant(C.Structure):
    _fields_ = [
        ("file", C.c_int64),
        ("offset", C.c_int64),
        ("src_file", C.c_int64),
        ("src_offset", C.c_int64),
        ("length", C.c_int64),
        ("amp", C.c_float),
        ("gain", C.c_float),
    ]


RING_RUN_DTYPE = np.dtype([("file", np.int32), ("dest", np.int32), ("src_file", np.int32), ("count", np.int32),
                           ("src_frame", np.int64)])

CANDIDATE_DTYPE = np.dtype(
    [("file", np.int32), ("offset", np.int32), ("sim", np.float32), ("boost", np.float32)]
)

_fp = C.POINTER(C.c_float)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol of include/strugatzki_amd.h
SIGNATURES = {
    "sgz_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "sgz_destroy": (None, [_vp]),
    "sgz_last_error": (C.c_char_p, [_vp]),
    "sgz_version": (C.c_char_p, []),
    "sgz_synchronize": (C.c_int, [_vp]),
    "sgz_db_reset": (C.c_int, [_vp, C.c_int]),
    "sgz_db_add_file": (C.c_int, [_vp, _vp, C.c_int64]),
    "sgz_db_add_file_pcm": (C.c_int, [_vp, _vp, C.c_int, C.c_int64]),
    "sgz_db_add_file_device": (C.c_int, [_vp, _vp, C.c_int64]),
    "sgz_db_add_synthetic": (C.c_int, [_vp, C.c_uint64, C.c_int64, C.c_int64, C.POINTER(Plant), C.c_int]),
    "sgz_db_num_files": (C.c_int, [_vp]),
    "sgz_db_file_frames": (C.c_int64, [_vp, C.c_int]),
    "sgz_db_read_file": (C.c_int, [_vp, C.c_int, _vp]),
    "sgz_db_read_frames": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int64, _vp]),
    "sgz_db_set_tag": (C.c_int, [_vp, C.c_char_p]),
    "sgz_db_tag": (C.c_char_p, [_vp]),
    "sgz_run_correlation_xml_multi": (C.c_int, [_vp, C.c_int, C.c_char_p, _vp, _vp, C.POINTER(_vp)]),
    "sgz_corr_ring_runs": (C.c_int64, [_vp, C.c_int, C.c_int, _vp, C.c_int64]),
    "sgz_corr_set_ring_image": (C.c_int, [_vp, C.c_int, _vp, C.c_int]),
    "sgz_norm_zero_frame": (C.c_int, [_vp, _vp]),
    "sgz_corr_search_multi": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int,
                                        C.c_int64, _vp, C.c_int, _vp]),
    "sgz_corr_search_batch": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                        C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, _vp,
                                        C.c_int, _vp]),
    "sgz_set_norms": (C.c_int, [_vp, _vp]),
    "sgz_corr_set_target": (C.c_int, [_vp, _vp, C.c_int, C.c_float, C.c_float]),
    "sgz_corr_set_target_device": (C.c_int, [_vp, _vp, C.c_int, C.c_float, C.c_float]),
    "sgz_corr_target_stats": (C.c_int, [_vp, _vp]),
    "sgz_corr_num_windows": (C.c_int64, [_vp, C.c_int]),
    "sgz_corr_set_tail_trim": (C.c_int, [_vp, C.c_int64]),
    "sgz_corr_run": (C.c_int, [_vp, C.c_int, C.c_int]),
    "sgz_corr_run_screen": (C.c_int, [_vp, C.c_int, C.c_int]),
    "sgz_get_stream": (C.c_void_p, [_vp]),
    "sgz_sel_feed_gathered": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, _vp, _vp, C.c_int, C.c_int]),
    "sgz_corr_search": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64,
                                  C.c_float, _vp, C.c_int, _vp]),
    "sgz_feature_stats": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "sgz_run_feature_stats": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    "sgz_write_norms_file": (C.c_int, [C.c_char_p, _vp, C.c_int]),
    "sgz_corr_run_screen_f16": (C.c_int, [_vp, C.c_int, C.c_int]),
    "sgz_corr_screen_f16_available": (C.c_int, [_vp]),
    "sgz_corr_set_screen_kernel": (C.c_int, [_vp, C.c_int]),
    "sgz_corr_screen_kernel_in_use": (C.c_int, [_vp]),
    "sgz_corr_screen_geometry": (C.c_int, [_vp, _vp, _vp, _vp]),
    "sgz_corr_set_spectra_format": (C.c_int, [_vp, C.c_int]),
    "sgz_corr_spectra_format_in_use": (C.c_int, [_vp]),
    "sgz_corr_set_screen_output": (C.c_int, [_vp, C.c_int]),
    "sgz_fft_reference_host": (C.c_int, [C.c_int, C.c_int, _vp, _vp]),
    "sgz_corr_read_dense": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "sgz_corr_read_dense_range": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "sgz_corr_compact": (C.c_int, [_vp, C.c_int, C.c_int, C.c_float, _vp, C.c_int64, C.POINTER(C.c_int64)]),
    "sgz_corr_compact_device": (C.c_int, [_vp, C.c_int, C.c_int, C.c_float, _vp, C.c_int64, _vp]),
    "sgz_corr_rescore_device": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "sgz_corr_rescore_device_noalloc": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "sgz_corr_compact_rescored": (C.c_int, [_vp, C.c_int, C.c_int, C.c_float, _vp, C.c_int64, C.POINTER(C.c_int64)]),
    "sgz_corr_compact_bounded": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int64, C.POINTER(C.c_int64)]),
    "sgz_corr_compact_bounded_device": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int64, _vp]),
    "sgz_corr_prefetch_dense": (C.c_int, [_vp, C.c_int, C.c_int]),
    "sgz_corr_compact_top": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int64, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_float)]),
    "sgz_corr_compact_top_device": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int64, _vp]),
    "sgz_sel_feed_top": (C.c_int, [_vp, _vp, C.c_int64, C.c_float]),
    "sgz_sel_feed_gathered_top": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, _vp, _vp]),
    "sgz_corr_last_kernel_ms": (C.c_float, [_vp]),
    "sgz_last_kernel_ms": (C.c_float, [_vp]),
    "sgz_kernel_ms_history": (C.c_int, [_vp, _vp, C.c_int]),
    "sgz_set_kernel_timing": (C.c_int, [_vp, C.c_int]),
    "sgz_corr_last_work": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "sgz_sel_create": (_vp, [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int]),
    "sgz_sel_destroy": (None, [_vp]),
    "sgz_sel_feed_file": (C.c_int, [_vp, C.c_int, C.c_int64, _vp, _vp, _vp]),
    "sgz_sel_feed_candidates": (C.c_int, [_vp, _vp, C.c_int64]),
    "sgz_sel_feed_file_punch": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp,
                                          _vp]),
    "sgz_sel_feed_screened": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int]),
    "sgz_sel_is_full": (C.c_int, [_vp]),
    "sgz_sel_threshold": (C.c_float, [_vp]),
    "sgz_sel_result": (C.c_int, [_vp, C.POINTER(Match), C.c_int]),
    "sgz_segm_curve": (
        C.c_int,
        [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_float, _vp, C.POINTER(C.c_int64)],
    ),
    "sgz_segm_select": (
        C.c_int,
        [_vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(Break), C.c_int],
    ),
    "sgz_selfsim_run": (
        C.c_int,
        [_vp, _vp, C.c_int64, _vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
         C.c_int, C.c_float, C.c_float, C.c_int, _vp, _vp],
    ),
    "sgz_selfsim_rows": (
        C.c_int,
        [_vp, _vp, C.c_int64, _vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
         C.c_int, C.c_int, _vp],
    ),
    "sgz_selfsim_release": (C.c_int, [_vp]),
    "sgz_multi_set_exchange": (C.c_int, [_vp, C.c_int]),
    "sgz_multi_last_exchange": (C.c_int, [_vp]),
    "sgz_run_selfsimilarity_rows_xml": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, _vp, _vp]),
    "sgz_extract_num_frames": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "sgz_extract": (
        C.c_int,
        [_vp, _vp, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int64,
         C.POINTER(C.c_int64)],
    ),
    "sgz_extract_pcm": (
        C.c_int,
        [_vp, _vp, C.c_int, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int64,
         C.POINTER(C.c_int64)],
    ),
    "sgz_run_extraction_xml": (C.c_int, [_vp, C.c_char_p, _vp, _vp]),
    "sgz_run_crosssimilarity_xml": (C.c_int, [_vp, C.c_char_p, _vp, _vp]),
    "sgz_run_correlation_xml": (C.c_int, [_vp, C.c_char_p, _vp, _vp, C.POINTER(_vp)]),
    "sgz_run_segmentation_xml": (C.c_int, [_vp, C.c_char_p, _vp, _vp, C.POINTER(_vp)]),
    "sgz_run_selfsimilarity_xml": (C.c_int, [_vp, C.c_char_p, _vp, _vp]),
    "sgz_host_last_error": (C.c_char_p, []),
    "sgz_free_string": (None, [_vp]),
    "sgz_config_roundtrip_xml": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_vp)]),
    "sgz_default_config_xml": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "sgz_write_float_aiff": (C.c_int, [C.c_char_p, _vp, C.c_int64, C.c_int, C.c_double]),
    "sgz_write_float_audio": (C.c_int, [C.c_char_p, C.c_char_p, _vp, C.c_int64, C.c_int, C.c_double]),
    "sgz_secs_to_frames": (C.c_int64, [C.c_double, C.c_double]),
    "sgz_correlate_config_from_seconds": (C.c_int, [C.c_char_p, C.c_char_p, C.c_double, C.c_double, C.c_double,
                                                    C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                                    C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int,
                                                    C.POINTER(_vp)]),
    "sgz_read_audio_file": (
        C.c_int,
        [C.c_char_p, _vp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_double)],
    ),
}

PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_double, C.c_void_p)


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C strugatzki_amd/csrc` "
            "(or __graft_entry__.build()); strugatzki_amd has no CPU fallback"
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class NativeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"strugatzki_amd error {code}: {message}")
        self.code = code


def ptr(a: np.ndarray | None):
    """Raw pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
