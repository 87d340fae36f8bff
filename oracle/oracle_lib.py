"""ctypes access to the CPU oracle (oracle/strugatzki_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg as the checker / reported baseline.  The product
package (strugatzki_amd) never imports this.  Parity unpinned: see the header of
strugatzki_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstrugatzki_oracle.so")


def build() -> None:
    subprocess.check_call(["make", "-C", _HERE, "-s"])


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


class CorrConfig(C.Structure):
    _fields_ = [
        ("punch_in_start", C.c_int64),
        ("punch_in_stop", C.c_int64),
        ("temporal_weight", C.c_float),
        ("normalize", C.c_int32),
        ("max_boost", C.c_float),
        ("num_matches", C.c_int32),
        ("num_per_file", C.c_int32),
        ("min_spacing", C.c_int64),
        ("step_size", C.c_int32),
        ("num_coeffs", C.c_int32),
    ]


class PunchConfig(C.Structure):
    """sgzo_punch_config: FeatureCorrelation with punchOut, minPunch, maxPunch."""
    _fields_ = [
        ("base", CorrConfig),
        ("punch_out_start", C.c_int64),
        ("punch_out_stop", C.c_int64),
        ("out_temporal_weight", C.c_float),
        ("pad_", C.c_int32),
        ("min_punch", C.c_int64),
        ("max_punch", C.c_int64),
    ]


class SegmConfig(C.Structure):
    _fields_ = [
        ("has_start", C.c_int32),
        ("has_stop", C.c_int32),
        ("span_start", C.c_int64),
        ("span_stop", C.c_int64),
        ("corr_len", C.c_int64),
        ("temporal_weight", C.c_float),
        ("normalize", C.c_int32),
        ("num_breaks", C.c_int32),
        ("min_spacing", C.c_int64),
        ("step_size", C.c_int32),
        ("num_coeffs", C.c_int32),
    ]


class SelfSimConfig(C.Structure):
    _fields_ = [
        ("has_start", C.c_int32),
        ("has_stop", C.c_int32),
        ("span_start", C.c_int64),
        ("span_stop", C.c_int64),
        ("corr_len", C.c_int64),
        ("decimation", C.c_int32),
        ("temporal_weight", C.c_float),
        ("colors", C.c_int32),
        ("color_warp", C.c_float),
        ("color_ceil", C.c_float),
        ("color_inv", C.c_int32),
        ("normalize", C.c_int32),
        ("step_size", C.c_int32),
        ("num_coeffs", C.c_int32),
    ]


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


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.sgzo_stat.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                C.POINTER(C.c_double)]
        L.sgzo_stat.restype = None
        L.sgzo_correlate.argtypes = [fp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, fp, C.c_int,
                                     C.c_double, C.c_double, C.c_int, C.c_int]
        L.sgzo_correlate.restype = C.c_float
        L.sgzo_correlate_half.argtypes = [C.c_int, C.c_int, fp, C.c_int, C.c_int]
        L.sgzo_correlate_half.restype = C.c_float
        L.sgzo_avg.argtypes = [fp, C.c_int, C.c_int]
        L.sgzo_avg.restype = C.c_float
        L.sgzo_normalize.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.sgzo_normalize.restype = None
        L.sgzo_full_to_feat.argtypes = [C.c_int64, C.c_int32]
        L.sgzo_full_to_feat.restype = C.c_int32
        L.sgzo_feat_to_full.argtypes = [C.c_int32, C.c_int32]
        L.sgzo_feat_to_full.restype = C.c_int64
        L.sgzo_correlation_run.argtypes = [C.POINTER(CorrConfig), fp, C.c_int64, fp, C.POINTER(fp),
                                           C.POINTER(C.c_int64), C.c_int, C.POINTER(fp), C.POINTER(fp),
                                           C.POINTER(Match)]
        L.sgzo_correlation_run.restype = C.c_int
        L.sgzo_correlation_num_windows.argtypes = [C.c_int64, C.c_int32]
        L.sgzo_correlation_num_windows.restype = C.c_int64
        L.sgzo_segmentation_run.argtypes = [C.POINTER(SegmConfig), fp, C.c_int64, fp, fp, C.POINTER(C.c_int64),
                                            C.POINTER(Break)]
        L.sgzo_segmentation_run.restype = C.c_int
        L.sgzo_selfsim_geometry.argtypes = [C.POINTER(SelfSimConfig), C.c_int64, C.c_int64, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.sgzo_selfsim_geometry.restype = C.c_int
        L.sgzo_selfsim_run.argtypes = [C.POINTER(SelfSimConfig), fp, C.c_int64, fp, C.c_int64, fp, fp,
                                       C.POINTER(C.c_int32)]
        L.sgzo_selfsim_run.restype = C.c_int
        L.sgzo_synth_base.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int]
        L.sgzo_synth_base.restype = C.c_float
        L.sgzo_synth_fill.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(Plant),
                                      C.c_int, fp]
        L.sgzo_synth_fill.restype = None
        L.sgzo_extract_num_frames.argtypes = [C.c_int64, C.c_int]
        L.sgzo_extract_num_frames.restype = C.c_int64
        L.sgzo_extract.argtypes = [fp, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.sgzo_extract.restype = C.c_int
        _lib = L
    return _lib


def _fp(a: np.ndarray | None):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def match_tuple(m) -> tuple:
    return (m.sim, m.file, m.start, m.stop, m.boost_in, m.boost_out)


# ---- MathUtil --------------------------------------------------------------
def stat(mat: np.ndarray, frame_off: int, frame_len: int, chan_off: int, chan_len: int) -> tuple[float, float]:
    mat = np.ascontiguousarray(mat, dtype=np.float32)
    m, s = C.c_double(), C.c_double()
    lib().sgzo_stat(_fp(mat), mat.shape[1], frame_off, frame_len, chan_off, chan_len, C.byref(m), C.byref(s))
    return m.value, s.value


def correlate(a: np.ndarray, a_mean: float, a_std: float, b: np.ndarray, b_mean: float, b_std: float,
              b_frame_off: int, b_chan_off: int) -> float:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return lib().sgzo_correlate(_fp(a), a.shape[1], a_mean, a_std, a.shape[1], a.shape[0], _fp(b), b.shape[1],
                                b_mean, b_std, b_frame_off, b_chan_off)


def correlate_half(num_channels: int, half: int, a: np.ndarray, frame_off: int, chan_off: int) -> float:
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.shape[1] == 2 * half
    return lib().sgzo_correlate_half(num_channels, half, _fp(a), frame_off, chan_off)


def avg(b: np.ndarray, off: int, length: int) -> float:
    b = np.ascontiguousarray(b, dtype=np.float32)
    return lib().sgzo_avg(_fp(b), off, length)


def normalize(norm: np.ndarray | None, b: np.ndarray, b_off: int, b_len: int) -> np.ndarray:
    b = np.array(b, dtype=np.float32, order="C")
    n = None if norm is None else np.ascontiguousarray(norm, dtype=np.float32)
    lib().sgzo_normalize(_fp(n), _fp(b), b.shape[0], b.shape[1], b_off, b_len)
    return b


def full_to_feat(n: int, step: int) -> int:
    return lib().sgzo_full_to_feat(n, step)


def num_windows(n: int, L: int) -> int:
    return lib().sgzo_correlation_num_windows(n, L)


# ---- drivers ---------------------------------------------------------------
def correlation_run(cfg: CorrConfig, target: np.ndarray, norms: np.ndarray | None, db: list[np.ndarray],
                    dense: bool = True):
    """Returns (matches, dense_sims, dense_boosts); dense lists are None if dense=False."""
    target = np.ascontiguousarray(target, dtype=np.float32)
    nrm = None if norms is None else np.ascontiguousarray(norms, dtype=np.float32)
    db = [np.ascontiguousarray(x, dtype=np.float32) for x in db]
    nf = len(db)
    L = full_to_feat(cfg.punch_in_stop, cfg.step_size) - full_to_feat(cfg.punch_in_start, cfg.step_size)
    fpp = C.POINTER(C.c_float)
    frames = (fpp * max(nf, 1))(*[_fp(x) for x in db])
    counts = (C.c_int64 * max(nf, 1))(*[x.shape[0] for x in db])
    sims = boosts = None
    sim_p = boost_p = None
    if dense:
        sims = [np.zeros(num_windows(x.shape[0], L), dtype=np.float32) for x in db]
        boosts = [np.zeros_like(s) for s in sims]
        sim_p = (fpp * max(nf, 1))(*[_fp(s) for s in sims])
        boost_p = (fpp * max(nf, 1))(*[_fp(s) for s in boosts])
    out = (Match * max(cfg.num_matches, 1))()
    n = lib().sgzo_correlation_run(C.byref(cfg), _fp(target), target.shape[0], _fp(nrm), frames, counts, nf,
                                   sim_p, boost_p, out)
    if n < 0:
        raise ValueError("oracle rejected the correlation arguments")
    return [out[i] for i in range(n)], sims, boosts


def correlation_run_punch(cfg: PunchConfig, target: np.ndarray, norms: np.ndarray | None, db: list[np.ndarray]):
    """Punch-in / punch-out search.  Returns (matches, dense_in, dense_out): per file [2][frames] arrays
    (row 0 sim, row 1 boost) indexed by window offset in the file, NaN where the reference computes nothing."""
    target = np.ascontiguousarray(target, dtype=np.float32)
    nrm = None if norms is None else np.ascontiguousarray(norms, dtype=np.float32)
    db = [np.ascontiguousarray(x, dtype=np.float32) for x in db]
    nf = len(db)
    fpp = C.POINTER(C.c_float)
    frames = (fpp * max(nf, 1))(*[_fp(x) for x in db])
    counts = (C.c_int64 * max(nf, 1))(*[x.shape[0] for x in db])
    din = [np.full((2, max(x.shape[0], 1)), np.nan, dtype=np.float32) for x in db]
    dout = [np.full((2, max(x.shape[0], 1)), np.nan, dtype=np.float32) for x in db]
    in_p = (fpp * max(nf, 1))(*[_fp(a) for a in din])
    out_p = (fpp * max(nf, 1))(*[_fp(a) for a in dout])
    out = (Match * max(cfg.base.num_matches, 1))()
    fn = lib().sgzo_correlation_run_punch
    fn.restype = C.c_int
    n = fn(C.byref(cfg), _fp(target), C.c_int64(target.shape[0]), _fp(nrm), frames, counts, C.c_int(nf), in_p, out_p,
           out)
    if n < 0:
        raise ValueError("oracle rejected the correlation arguments")
    return [out[i] for i in range(n)], din, dout


def crosssim_span(span_start, span_stop, num_frames: int, step: int) -> tuple[int, int]:
    """(first frame, frame count) of a span (None = open end) in a feature file; CrossSimilarityImpl.openInput."""
    a, n = C.c_int64(), C.c_int64()
    lib().sgzo_crosssim_span(C.c_int(span_start is not None), C.c_int(span_stop is not None),
                             C.c_int64(span_start or 0), C.c_int64(span_stop or 0), C.c_int64(num_frames),
                             C.c_int(step), C.byref(a), C.byref(n))
    return a.value, n.value


def crosssim_run(frames1: np.ndarray, frames2: np.ndarray, norms: np.ndarray | None, weight: float,
                 max_boost: float) -> np.ndarray:
    """Similarity curve of the shorter array slid along the longer one (CrossSimilarity)."""
    f1 = np.ascontiguousarray(frames1, dtype=np.float32)
    f2 = np.ascontiguousarray(frames2, dtype=np.float32)
    nrm = None if norms is None else np.ascontiguousarray(norms, dtype=np.float32)
    out = np.zeros(max(f1.shape[0], f2.shape[0], 1), dtype=np.float32)
    fn = lib().sgzo_crosssim_run
    fn.restype = C.c_int64
    n = fn(_fp(f1), C.c_int64(f1.shape[0]), _fp(f2), C.c_int64(f2.shape[0]), C.c_int(f1.shape[1]), _fp(nrm),
           C.c_float(weight), C.c_float(max_boost), _fp(out))
    if n < 0:
        raise ValueError("oracle rejected the cross-similarity arguments")
    return out[:n].copy()


def feature_stats(files: list[np.ndarray]) -> np.ndarray:
    """FeatureStats over feature files ([frames][numCh] each): [numCh][2] doubles (1st, 99th percentile)."""
    files = [np.ascontiguousarray(x, dtype=np.float32) for x in files]
    nch = files[0].shape[1]
    fpp = C.POINTER(C.c_float)
    ptrs = (fpp * len(files))(*[_fp(x) for x in files])
    counts = (C.c_int64 * len(files))(*[x.shape[0] for x in files])
    out = np.zeros((nch, 2), dtype=np.float64)
    fn = lib().sgzo_feature_stats
    fn.restype = C.c_int
    rc = fn(ptrs, counts, C.c_int(len(files)), C.c_int(nch), out.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise ValueError("oracle rejected the feature statistics arguments")
    return out


def segmentation_run(cfg: SegmConfig, frames: np.ndarray, norms: np.ndarray | None):
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    nrm = None if norms is None else np.ascontiguousarray(norms, dtype=np.float32)
    dense = np.zeros(max(frames.shape[0], 1), dtype=np.float32)
    nd = C.c_int64(0)
    out = (Break * max(cfg.num_breaks, 1))()
    n = lib().sgzo_segmentation_run(C.byref(cfg), _fp(frames), frames.shape[0], _fp(nrm), _fp(dense),
                                    C.byref(nd), out)
    if n < 0:
        raise ValueError("oracle rejected the segmentation arguments")
    return [out[i] for i in range(n)], dense[: nd.value].copy()


def selfsim_geometry(cfg: SelfSimConfig, n1: int, n2: int) -> tuple[int, int, int, int]:
    d, a, c = C.c_int32(), C.c_int32(), C.c_int32()
    ext = lib().sgzo_selfsim_geometry(C.byref(cfg), n1, n2, C.byref(d), C.byref(a), C.byref(c))
    return ext, d.value, a.value, c.value


def selfsim_run(cfg: SelfSimConfig, frames1: np.ndarray, frames2: np.ndarray | None, norms: np.ndarray | None,
                want_img: bool = True):
    f1 = np.ascontiguousarray(frames1, dtype=np.float32)
    f2 = f1 if frames2 is None else np.ascontiguousarray(frames2, dtype=np.float32)
    nrm = None if norms is None else np.ascontiguousarray(norms, dtype=np.float32)
    ext, _, _, _ = selfsim_geometry(cfg, f1.shape[0], f2.shape[0])
    if ext < 0:
        raise ValueError("illegal self-similarity geometry")
    sim = np.zeros((ext, ext), dtype=np.float32)
    img = np.zeros((ext, ext), dtype=np.int32) if want_img else None
    rc = lib().sgzo_selfsim_run(C.byref(cfg), _fp(f1), f1.shape[0], _fp(f2), f2.shape[0], _fp(nrm), _fp(sim),
                                None if img is None else img.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc < 0:
        raise ValueError("oracle rejected the self-similarity arguments")
    return sim, img


# ---- synthetic data --------------------------------------------------------
def synth_file(seed: int, file: int, num_frames: int, num_ch: int, plants: list[Plant] = (), frame0: int = 0
               ) -> np.ndarray:
    out = np.empty((num_frames, num_ch), dtype=np.float32)
    arr = (Plant * max(len(plants), 1))(*plants)
    lib().sgzo_synth_fill(seed, file, frame0, num_frames, num_ch, arr, len(plants), _fp(out))
    return out


def _mix64(z: np.ndarray) -> np.ndarray:
    z = z + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synth_file_numpy(seed: int, file: int, num_frames: int, num_ch: int) -> np.ndarray:
    """numpy twin of sgzo_synth_fill without plants (cross-checks the C and HIP generators)."""
    with np.errstate(over="ignore"):
        k = _mix64(np.uint64(seed) + np.uint64(file % (1 << 64)) * np.uint64(0xD1342543DE82EF95))
        fr = np.arange(num_frames + 8, dtype=np.uint64)[:, None]
        ch = np.arange(num_ch, dtype=np.uint64)[None, :]
        h = _mix64(k ^ (fr * np.uint64(16) + ch))
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    s = np.zeros((num_frames, num_ch), dtype=np.float32)
    for j in range(8):
        s = s + u[j: j + num_frames]
    s = s * np.float32(0.125)
    out = s.copy()
    out[:, 0] = np.float32(0.05) + np.float32(1.45) * s[:, 0]
    return out


# ---- extraction (own spectral model; framing per the reference) ----------------
def extract(pcm: np.ndarray, sample_rate: float, fft_size: int = 1024, fft_overlap: int = 2, num_coeffs: int = 13,
            channels: int = 0) -> np.ndarray:
    a = np.ascontiguousarray(pcm, dtype=np.float32)
    if a.ndim == 1:
        a = a[:, None]
    n = lib().sgzo_extract_num_frames(a.shape[0], fft_size // fft_overlap)
    out = np.zeros((n, num_coeffs + 1), dtype=np.float32)
    rc = lib().sgzo_extract(_fp(a), a.shape[0], a.shape[1], sample_rate, fft_size, fft_overlap, num_coeffs, channels,
                            _fp(out))
    if rc != 0:
        raise ValueError("oracle rejected the extraction arguments")
    return out
