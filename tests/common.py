"""Shared builders for the tests: synthetic feature databases and configs."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle_lib as O  # noqa: E402

SEED = 0x5354525547  # "STRUG"
NUM_CH = 14
STEP = 512
TARGET_FILE = -1


def norms_default(num_ch: int = NUM_CH) -> np.ndarray:
    """[numCh][2] (min, max): loudness 0..1.5, MFCC channels slightly different ranges each."""
    n = np.zeros((num_ch, 2), dtype=np.float32)
    n[0] = (0.0, 1.5)
    for c in range(1, num_ch):
        n[c] = (np.float32(0.01) * c, np.float32(1.0) + np.float32(0.02) * c)
    return n


def make_plants(num: int, L: int, num_files: int, frames_per_file: int, target_start: int, amp_step: float = 0.02,
                first_file: int = 0):
    """`num` noisy copies of the target, copy j in file first_file + j*stride at a distinct offset."""
    plants = []
    stride = max(1, (num_files - first_file) // max(num, 1))
    for j in range(num):
        f = first_file + j * stride
        room = frames_per_file - L
        off = (37 + 101 * j) % max(room, 1) if room > 0 else 0
        plants.append(dict(file=f, offset=off, src_file=TARGET_FILE, src_offset=target_start, length=L,
                           amp=float(np.float32(amp_step) * (j + 1)), gain=1.0))
    return plants


def oracle_plants(plants):
    return [O.Plant(p["file"], p["offset"], p["src_file"], p["src_offset"], p["length"], p["amp"], p["gain"])
            for p in plants]


def native_plants(plants):
    from strugatzki_amd import _native as N
    return [N.Plant(p["file"], p["offset"], p["src_file"], p["src_offset"], p["length"], p["amp"], p["gain"])
            for p in plants]


def synth_db(num_files: int, frames, plants=(), seed: int = SEED, num_ch: int = NUM_CH):
    """Host copy of the synthetic database (list of [frames][numCh] arrays)."""
    if np.isscalar(frames):
        frames = [int(frames)] * num_files
    op = oracle_plants(plants)
    return [O.synth_file(seed, f, int(frames[f]), num_ch, op) for f in range(num_files)]


def synth_target(target_start: int, L: int, seed: int = SEED, num_ch: int = NUM_CH) -> np.ndarray:
    """Frames [0, target_start + L + 16) of the dedicated target file."""
    return O.synth_file(seed, TARGET_FILE, target_start + L + 16, num_ch)


def punch_config(in_start: int, in_len: int, out_start: int, out_len: int, *, min_punch: int, max_punch: int,
                 out_weight=0.5, **kw) -> "O.PunchConfig":
    """Config of a punch-in / punch-out search; spans and punch lengths in feature frames."""
    base = corr_config(in_start, in_len, **kw)
    return O.PunchConfig(base, out_start * STEP, (out_start + out_len) * STEP, out_weight, 0, min_punch * STEP,
                         max_punch * STEP)


def corr_config(target_start: int, L: int, *, weight=0.5, normalize=True, max_boost=8.0, num_matches=10,
                num_per_file=1, min_spacing=0, num_ch: int = NUM_CH) -> O.CorrConfig:
    return O.CorrConfig(target_start * STEP, (target_start + L) * STEP, weight, int(normalize), max_boost,
                        num_matches, num_per_file, min_spacing, STEP, num_ch - 1)


SHARDED_CASE = dict(L=96, ts=17, nf=14, frames=900, num_matches=4, nplants=5)


def sharded_case_db(mode: str):
    """Database of the fixed sharded-search case (tests/sharded_rank.py and its callers).  Mode "nan": file 2
    starts with a constant stretch (its first windows have no spread: NaN similarities, the first of which the
    reference's loop keeps while the sets have room), file 5 holds one further inside."""
    q = SHARDED_CASE
    db = synth_db(q["nf"], q["frames"], make_plants(q["nplants"], q["L"], q["nf"], q["frames"], q["ts"]))
    if mode == "nan":
        db[2][0:300] = db[2][0]
        db[5][400:700] = db[5][400]
    return db


class HipMem:
    """Device memory for a test from the HIP runtime the library itself is linked to: the symbols are looked
    up through the library's handle (a process that has also loaded torch holds a second copy of the runtime,
    which knows nothing of the library's device state)."""

    def __init__(self):
        import ctypes as C
        from strugatzki_amd import _native as N
        self.C = C
        self.lib = N.lib
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.bufs = []

    def free_bytes(self):
        free, total = self.C.c_size_t(), self.C.c_size_t()
        assert self.lib.hipMemGetInfo(self.C.byref(free), self.C.byref(total)) == 0
        return free.value

    def alloc(self, nbytes):
        p = self.C.c_void_p()
        assert self.lib.hipMalloc(self.C.byref(p), max(nbytes, 16)) == 0
        self.bufs.append(p)
        return p.value

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        assert self.lib.hipMemcpy(dptr, arr.ctypes.data, arr.nbytes, 1) == 0  # hipMemcpyHostToDevice

    def download(self, dptr, arr):
        assert self.lib.hipMemcpy(arr.ctypes.data, dptr, arr.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return arr

    def free_all(self):
        for p in self.bufs:
            self.lib.hipFree(p)
        self.bufs = []
