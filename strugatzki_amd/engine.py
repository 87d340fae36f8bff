"""Thin object wrapper over the C ABI: one `Context` per GPU / processor.

Everything numeric happens inside libstrugatzki_amd.so; this module only moves
pointers and raises on error codes.
"""
from __future__ import annotations

import ctypes as C
import sys
from typing import Iterable, Sequence

import numpy as np

from . import _native as N

lib = N.lib


def _as_frames(a: np.ndarray, num_ch: int | None = None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("frames must be a [numFrames][numCh] array")
    if num_ch is not None and a.shape[1] != num_ch:
        raise ValueError(f"expected {num_ch} channels, got {a.shape[1]}")
    return a


_KEEP_STREAMS: list = []  # torch streams handed to contexts: never collected before the process ends


class Context:
    """Owns one sgz_context (GPU `device`, optional existing HIP stream handle)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        h = C.c_void_p()
        rc = lib.sgz_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != N.SGZ_OK:
            raise N.NativeError(rc, "sgz_create failed (no usable GPU?)")
        self._h = h
        self.num_channels = 0
        self.torch_stream = None

    @classmethod
    def on_torch_stream(cls, device: int = 0) -> "Context":
        """A context whose kernels run on a stream torch owns (and that stays alive until the process ends).
        Work torch queues under `torch.cuda.stream(ctx.torch_stream)` -- an RCCL collective, a copy -- is then
        ordered after the library's kernels on the device, without a host-side wait in between."""
        import torch

        ts = torch.cuda.Stream(device=device)
        ctx = cls(device, ts.cuda_stream)
        ctx.torch_stream = ts
        _KEEP_STREAMS.append(ts)
        return ctx

    def close(self) -> None:
        if self._h:
            lib.sgz_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != N.SGZ_OK:
            raise N.NativeError(rc, lib.sgz_last_error(self._h).decode())

    # ---- database --------------------------------------------------------
    def db_reset(self, num_channels: int) -> None:
        self._check(lib.sgz_db_reset(self._h, num_channels))
        self.num_channels = num_channels
        # the sharded search keeps every rank's file placement between searches: a new database starts afresh
        # (when the files of one rank change, every rank of the group has to go through db_reset)
        self.__dict__.pop("_rank_maps", None)
        self.__dict__.pop("_placement", None)

    def db_add_file(self, frames: np.ndarray) -> None:
        a = _as_frames(frames, self.num_channels)
        self._check(lib.sgz_db_add_file(self._h, N.ptr(a), a.shape[0]))

    def db_add_file_pcm(self, samples, sample_format: int, num_frames: int) -> None:
        """Frames as an audio file holds them (bytes or a uint8 array, interleaved, layout SGZ_PCM_*)."""
        raw = np.frombuffer(samples, dtype=np.uint8) if not isinstance(samples, np.ndarray) else samples
        raw = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        width = {0: 4, 1: 4, 2: 2, 3: 2, 4: 3, 5: 3, 6: 4, 7: 4}.get(sample_format, 0)
        if raw.size < width * num_frames * self.num_channels:
            raise ValueError("fewer bytes than num_frames * num_channels samples")
        self._check(lib.sgz_db_add_file_pcm(self._h, N.ptr(raw), sample_format, num_frames))

    def db_add_file_device(self, device_ptr: int, num_frames: int) -> None:
        self._check(lib.sgz_db_add_file_device(self._h, C.c_void_p(device_ptr), num_frames))

    def db_add_synthetic(self, seed: int, file_id: int, num_frames: int, plants: Sequence[N.Plant] = ()) -> None:
        arr = (N.Plant * len(plants))(*plants) if plants else None
        self._check(lib.sgz_db_add_synthetic(self._h, seed, file_id, num_frames, arr, len(plants)))

    @property
    def num_files(self) -> int:
        return lib.sgz_db_num_files(self._h)

    def file_frames(self, file: int) -> int:
        return lib.sgz_db_file_frames(self._h, file)

    def db_read_file(self, file: int) -> np.ndarray:
        n = self.file_frames(file)
        out = np.empty((n, self.num_channels), dtype=np.float32)
        self._check(lib.sgz_db_read_file(self._h, file, N.ptr(out)))
        return out

    def db_read_frames(self, file: int, first: int, count: int) -> np.ndarray:
        out = np.empty((count, self.num_channels), dtype=np.float32)
        self._check(lib.sgz_db_read_frames(self._h, file, first, count, N.ptr(out)))
        return out

    def corr_set_ring_image(self, file: int, frames: np.ndarray | None) -> None:
        """Ring contents behind the own frames of a file shorter than the target (sgz_corr_set_ring_image)."""
        if frames is None:
            self._check(lib.sgz_corr_set_ring_image(self._h, file, None, 0))
            return
        a = _as_frames(frames, self.num_channels)
        self._check(lib.sgz_corr_set_ring_image(self._h, file, N.ptr(a), a.shape[0]))

    def norm_zero_frame(self) -> np.ndarray:
        out = np.zeros(self.num_channels, dtype=np.float32)
        self._check(lib.sgz_norm_zero_frame(self._h, N.ptr(out)))
        return out

    def set_norms(self, norms: np.ndarray | None) -> None:
        """norms: [numCh][2] (min, max) or None."""
        if norms is None:
            self._check(lib.sgz_set_norms(self._h, None))
            self._norms_key = None
            return
        a = np.ascontiguousarray(norms, dtype=np.float32)
        if a.shape != (self.num_channels, 2):
            raise ValueError("norms must be [numCh][2]")
        self._check(lib.sgz_set_norms(self._h, N.ptr(a)))
        self._norms_key = a.tobytes()  # (the sharded search keys what it derived from the norms on this)

    # ---- correlation -----------------------------------------------------
    def corr_set_target(self, frames: np.ndarray, temporal_weight: float, max_boost: float) -> None:
        a = _as_frames(frames, self.num_channels)
        self._check(lib.sgz_corr_set_target(self._h, N.ptr(a), a.shape[0], temporal_weight, max_boost))
        self.target_frames = a.shape[0]

    def corr_set_target_device(self, device_ptr: int, num_frames: int, temporal_weight: float, max_boost: float) -> None:
        """corr_set_target for raw frames [num_frames][channels] (fp32) that are in the GPU's memory already
        (sgz_corr_set_target_device): `device_ptr` is their address there."""
        self._check(lib.sgz_corr_set_target_device(self._h, C.c_void_p(device_ptr), num_frames, temporal_weight,
                                                   max_boost))
        self.target_frames = num_frames

    def corr_target_stats(self) -> np.ndarray:
        """(mean, std-dev) of the loudness row, (mean, std-dev) of the other rows, ln(average loudness) of the target."""
        out = np.zeros(5, dtype=np.float64)
        self._check(lib.sgz_corr_target_stats(self._h, N.ptr(out)))
        return out

    def corr_num_windows(self, file: int) -> int:
        return lib.sgz_corr_num_windows(self._h, file)

    def corr_set_tail_trim(self, frames: int) -> None:
        """The passes that follow stop `frames` frames before the end of every file (the punch-in pass of a
        search with punch-out, impl/FeatureCorrelationImpl.scala:182-185); 0 = whole files."""
        self._check(lib.sgz_corr_set_tail_trim(self._h, int(frames)))

    def corr_run(self, file_begin: int = 0, file_end: int | None = None) -> None:
        if file_end is None:
            file_end = self.num_files
        self._check(lib.sgz_corr_run(self._h, file_begin, file_end))

    def corr_run_screen(self, file_begin: int = 0, file_end: int | None = None) -> None:
        if file_end is None:
            file_end = self.num_files
        self._check(lib.sgz_corr_run_screen(self._h, file_begin, file_end))

    def corr_run_screen_f16(self, file_begin: int = 0, file_end: int | None = None) -> None:
        if file_end is None:
            file_end = self.num_files
        self._check(lib.sgz_corr_run_screen_f16(self._h, file_begin, file_end))

    @property
    def screen_f16_available(self) -> bool:
        return bool(lib.sgz_corr_screen_f16_available(self._h))

    SCREEN_KERNELS = {"auto": 0, "matrix": 1, "spectral": 2}

    def set_screen_kernel(self, kernel: str) -> None:
        """Which kernel produces the bounds of corr_run_screen_f16: "matrix" (half-precision products on the
        matrix cores), "spectral" (fp32 transforms of whole blocks, HBM-bound) or "auto"."""
        self._check(lib.sgz_corr_set_screen_kernel(self._h, self.SCREEN_KERNELS[kernel]))

    def set_screen_output(self, lists_only: bool) -> None:
        """True: a run of the spectral kernel keeps only the list of the windows that can hold their file's maximum
        (what corr_compact_top* / corr_compact_bounded* need); dense reads of such a run fail."""
        self._check(lib.sgz_corr_set_screen_output(self._h, 1 if lists_only else 0))

    @property
    def screen_kernel_in_use(self) -> str | None:
        return {1: "matrix", 2: "spectral", 0: "none"}.get(lib.sgz_corr_screen_kernel_in_use(self._h))

    @property
    def screen_geometry(self) -> tuple[int, int, int]:
        """(N, windows per block, blocks) of the spectral kernel for the current layout; zeros when not in use."""
        n, hop, blocks = C.c_int(0), C.c_int(0), C.c_longlong(0)
        self._check(lib.sgz_corr_screen_geometry(self._h, C.byref(n), C.byref(hop), C.byref(blocks)))
        return n.value, hop.value, blocks.value

    def set_spectra_format(self, fmt: str) -> None:
        """'auto', 'f32' or 'q16': how the spectral kernel keeps the database's spectra."""
        self._check(lib.sgz_corr_set_spectra_format(self._h, {"auto": 0, "f32": 1, "q16": 2}[fmt]))

    @property
    def spectra_format_in_use(self) -> str | None:
        return {1: "f32", 2: "q16", 0: "none"}.get(lib.sgz_corr_spectra_format_in_use(self._h))

    def corr_read_dense(self, file: int) -> tuple[np.ndarray, np.ndarray]:
        n = max(0, self.corr_num_windows(file))
        sim = np.empty(n, dtype=np.float32)
        boost = np.empty(n, dtype=np.float32)
        self._check(lib.sgz_corr_read_dense(self._h, file, N.ptr(sim), N.ptr(boost)))
        return sim, boost

    def corr_prefetch_dense(self, file_begin: int, file_end: int) -> None:
        """Starts the transfer of the dense results of these files to the host, in order behind the runs queued so
        far; the next corr_read_dense_range over exactly these files picks it up (sgz_corr_prefetch_dense)."""
        self._check(lib.sgz_corr_prefetch_dense(self._h, file_begin, file_end))

    def corr_read_dense_range(self, file_begin: int, file_end: int) -> list[tuple[np.ndarray, np.ndarray]]:
        """(sim, boost) of every file in [file_begin, file_end), fetched with one pair of copies."""
        counts = [max(0, self.corr_num_windows(f)) for f in range(file_begin, file_end)]
        total = int(sum(counts))
        sim = np.empty(total, dtype=np.float32)
        boost = np.empty(total, dtype=np.float32)
        self._check(lib.sgz_corr_read_dense_range(self._h, file_begin, file_end, N.ptr(sim), N.ptr(boost)))
        out, pos = [], 0
        for n in counts:
            out.append((sim[pos:pos + n], boost[pos:pos + n]))
            pos += n
        return out

    def corr_compact(self, file_begin: int, file_end: int, threshold: float, capacity: int,
                     rescore: bool = False) -> np.ndarray:
        """Candidates with sim > threshold, sorted by (file, offset); raises on overflow.
        rescore=True replaces their sim / boost by the exact fp64 re-evaluation."""
        out = np.empty(max(capacity, 1), dtype=N.CANDIDATE_DTYPE)
        cnt = C.c_int64(0)
        fn = lib.sgz_corr_compact_rescored if rescore else lib.sgz_corr_compact
        rc = fn(self._h, file_begin, file_end, threshold, N.ptr(out), capacity, C.byref(cnt))
        self._check(rc)
        return out[: cnt.value]

    def corr_compact_bounded(self, file_begin: int, file_end: int, head_end: int, capacity: int) -> np.ndarray:
        """Exactly re-scored candidates of a search over the bounds of a corr_run_screen_f16 run
        (numPerFile = 1), sorted by (file, offset); raises NativeError(SGZ_E_OVERFLOW) when the exact path
        has to be taken."""
        out = np.empty(max(capacity, 1), dtype=N.CANDIDATE_DTYPE)
        cnt = C.c_int64(0)
        self._check(lib.sgz_corr_compact_bounded(self._h, file_begin, file_end, head_end, N.ptr(out), capacity,
                                                 C.byref(cnt)))
        return out[: cnt.value]

    def corr_compact_top(self, file_begin: int, file_end: int, num_top: int, capacity: int) -> tuple[np.ndarray, float]:
        """corr_compact_bounded over all files as head files, without the files that cannot reach the num_top-th
        largest of the files' lower bounds; returns the candidates and that bound (sgz_corr_compact_top)."""
        out = np.empty(max(capacity, 1), dtype=N.CANDIDATE_DTYPE)
        cnt = C.c_int64(0)
        thr = C.c_float(0.0)
        self._check(lib.sgz_corr_compact_top(self._h, file_begin, file_end, num_top, N.ptr(out), capacity,
                                             C.byref(cnt), C.byref(thr)))
        return out[: cnt.value], float(thr.value)

    def corr_compact_top_device(self, file_begin: int, file_end: int, num_top: int, out_ptr: int, capacity: int,
                                header_ptr: int) -> None:
        self._check(lib.sgz_corr_compact_top_device(self._h, file_begin, file_end, num_top, C.c_void_p(out_ptr),
                                                    capacity, C.c_void_p(header_ptr)))

    def corr_search(self, num_matches: int, num_per_file: int, min_spacing: int, step_size: int,
                    screen_mode: int = N.SGZ_SCREEN_AUTO, over_bounds: bool = True, capacity: int = 1 << 16,
                    margin: float = -1.0) -> list[N.Match]:
        """One whole search over the files of the context with the current target (sgz_corr_search)."""
        arr = (N.Match * max(num_matches, 1))()
        cnt = C.c_int(0)
        self._check(lib.sgz_corr_search(self._h, self.target_frames, num_matches, num_per_file, min_spacing,
                                        step_size, screen_mode, int(over_bounds), capacity, margin, arr,
                                        num_matches, C.byref(cnt)))
        return [arr[i] for i in range(cnt.value)]

    def corr_compact_bounded_device(self, file_begin: int, file_end: int, head_end: int, out_ptr: int,
                                    capacity: int, count_ptr: int) -> None:
        self._check(lib.sgz_corr_compact_bounded_device(self._h, file_begin, file_end, head_end,
                                                        C.c_void_p(out_ptr), capacity, C.c_void_p(count_ptr)))

    def corr_compact_device(self, file_begin: int, file_end: int, threshold: float, out_ptr: int,
                            capacity: int, count_ptr: int) -> None:
        self._check(lib.sgz_corr_compact_device(self._h, file_begin, file_end, threshold,
                                                C.c_void_p(out_ptr), capacity, C.c_void_p(count_ptr)))

    def corr_rescore_device(self, cands_ptr: int, capacity: int, count_ptr: int) -> None:
        self._check(lib.sgz_corr_rescore_device(self._h, C.c_void_p(cands_ptr), capacity, C.c_void_p(count_ptr)))

    def corr_rescore_device_noalloc(self, cands_ptr: int, capacity: int, count_ptr: int) -> None:
        """corr_rescore_device without any device allocation (fixed scratch array inside the library)."""
        self._check(lib.sgz_corr_rescore_device_noalloc(self._h, C.c_void_p(cands_ptr), capacity,
                                                        C.c_void_p(count_ptr)))

    def corr_last_kernel_ms(self) -> float:
        return float(lib.sgz_corr_last_kernel_ms(self._h))

    def set_kernel_timing(self, every: int) -> None:
        """Timing events with every run (1, the default), every n-th run (n) or none (0)."""
        self._check(lib.sgz_set_kernel_timing(self._h, every))

    def kernel_ms_history(self, n: int) -> np.ndarray:
        """Kernel times (ms) of the latest n runs, oldest first (HIP events recorded with each run)."""
        out = np.zeros(max(n, 1), dtype=np.float32)
        got = lib.sgz_kernel_ms_history(self._h, N.ptr(out), n)
        if got < 0:
            self._check(got)
        return out[:got]

    def corr_last_work(self) -> tuple[int, float]:
        w = C.c_int64(0)
        f = C.c_double(0.0)
        self._check(lib.sgz_corr_last_work(self._h, C.byref(w), C.byref(f)))
        return w.value, f.value

    @property
    def stream_handle(self) -> int:
        """hipStream_t of the context (0 = the null stream), e.g. for torch.cuda.ExternalStream."""
        return int(lib.sgz_get_stream(self._h) or 0)

    def synchronize(self) -> None:
        self._check(lib.sgz_synchronize(self._h))

    # ---- feature statistics ------------------------------------------------
    def feature_stats(self, file_begin: int = 0, file_end: int | None = None, per_file: bool = False):
        """[numCh][2] doubles: smallest 1st / largest 99th percentile over the files (FeatureStats); with
        per_file=True also the [files][numCh][2] percentiles of every file."""
        if file_end is None:
            file_end = self.num_files
        out = np.zeros((self.num_channels, 2), dtype=np.float64)
        pf = np.zeros((max(file_end - file_begin, 1), self.num_channels, 2), dtype=np.float64) if per_file else None
        self._check(lib.sgz_feature_stats(self._h, file_begin, file_end, N.ptr(out), N.ptr(pf) if per_file else None))
        return (out, pf) if per_file else out

    # ---- segmentation / self-similarity ----------------------------------
    def segm_curve(self, frames: np.ndarray, norms: np.ndarray | None, af_start: int, af_stop: int, half_win: int,
                   temporal_weight: float) -> np.ndarray:
        a = _as_frames(frames)
        nrm = None if norms is None else np.ascontiguousarray(norms, dtype=np.float32)
        n = max(0, (af_stop - af_start) - 2 * half_win + 1)  # (a short non-empty span: 1 value)
        out = np.empty(max(n, 1), dtype=np.float32)
        cnt = C.c_int64(0)
        self._check(lib.sgz_segm_curve(self._h, N.ptr(a), a.shape[0], a.shape[1], N.ptr(nrm), af_start, af_stop, half_win,
                                       temporal_weight, N.ptr(out), C.byref(cnt)))
        return out[: cnt.value]

    def selfsim(self, frames1: np.ndarray, frames2: np.ndarray | None, norms: np.ndarray | None, af_start: int,
                half_win: int, decim: int,
                img_ext: int, temporal_weight: float, colors: int = 0, color_warp: float = 1.0,
                color_ceil: float = 1.0, color_inv: bool = False, want_sim: bool = True,
                want_img: bool = True) -> tuple[np.ndarray | None, np.ndarray | None]:
        a1 = _as_frames(frames1)
        a2 = a1 if frames2 is None else _as_frames(frames2, a1.shape[1])
        nrm = None if norms is None else np.ascontiguousarray(norms, dtype=np.float32)
        sim = np.zeros((img_ext, img_ext), dtype=np.float32) if want_sim else None
        img = np.zeros((img_ext, img_ext), dtype=np.int32) if want_img else None
        self._check(lib.sgz_selfsim_run(self._h, N.ptr(a1), a1.shape[0], N.ptr(a2), a2.shape[0], a1.shape[1],
                                        N.ptr(nrm), af_start, half_win, decim, img_ext, temporal_weight, colors, color_warp,
                                        color_ceil, int(color_inv), N.ptr(sim), N.ptr(img)))
        return sim, img

    def selfsim_rows(self, frames1: np.ndarray, frames2: np.ndarray | None, norms: np.ndarray | None, af_start: int,
                     half_win: int, decim: int, img_ext: int, temporal_weight: float, x_begin: int,
                     x_end: int) -> np.ndarray:
        """Rows [x_begin, x_end) of the similarity matrix (sgz_selfsim_rows): the matrix stays on the GPU between
        calls with the same inputs."""
        a1 = _as_frames(frames1)
        a2 = a1 if frames2 is None else _as_frames(frames2, a1.shape[1])
        nrm = None if norms is None else np.ascontiguousarray(norms, dtype=np.float32)
        out = np.zeros((max(x_end - x_begin, 0), img_ext), dtype=np.float32)
        self._check(lib.sgz_selfsim_rows(self._h, N.ptr(a1), a1.shape[0], N.ptr(a2), a2.shape[0], a1.shape[1],
                                         N.ptr(nrm), af_start, half_win, decim, img_ext, temporal_weight, x_begin, x_end,
                                         N.ptr(out) if out.size else None))
        return out

    def selfsim_release(self) -> None:
        self._check(lib.sgz_selfsim_release(self._h))

    # ---- extraction -----------------------------------------------------------
    # sample layouts of sgz_extract_pcm (SGZ_PCM_*): (bytes per sample, little-endian code); big endian = code + 1
    PCM_F32, PCM_S16, PCM_S24, PCM_S32 = 0, 2, 4, 6

    def extract(self, pcm: np.ndarray, sample_rate: float, fft_size: int = 1024, fft_overlap: int = 2,
                num_coeffs: int = 13, channels_behavior: int = 0) -> np.ndarray:
        """pcm: [frames][channels] (or 1-D mono) -> [featureFrames][num_coeffs + 1].

        int16 / int32 arrays and float32 arrays of either byte order are uploaded as they are and decoded on the
        device (an integer of b bits counts as v / 2^(b-1), as audio files hold them); anything else is converted
        to float32 on the host first."""
        a = np.asarray(pcm)
        kind = (a.dtype.kind, a.dtype.itemsize)
        if kind in (("i", 2), ("i", 4), ("f", 4)):
            fmt = {("i", 2): self.PCM_S16, ("i", 4): self.PCM_S32, ("f", 4): self.PCM_F32}[kind]
            fmt += 0 if a.dtype.isnative == (sys.byteorder == "little") else 1
            a = np.ascontiguousarray(a)
        else:
            fmt = self.PCM_F32
            a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim == 1:
            a = a[:, None]
        return self._extract(a, fmt, a.shape[0], a.shape[1], sample_rate, fft_size, fft_overlap, num_coeffs,
                             channels_behavior)

    def extract_pcm(self, samples, sample_format: int, num_frames: int, num_channels: int, sample_rate: float,
                    fft_size: int = 1024, fft_overlap: int = 2, num_coeffs: int = 13,
                    channels_behavior: int = 0) -> np.ndarray:
        """The samples as an audio file holds them (bytes or a uint8 array, interleaved, layout SGZ_PCM_*)."""
        raw = np.frombuffer(samples, dtype=np.uint8) if not isinstance(samples, np.ndarray) else samples
        raw = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        width = {0: 4, 1: 4, 2: 2, 3: 2, 4: 3, 5: 3, 6: 4, 7: 4}.get(sample_format, 0)  # (unknown: the library says so)
        if raw.size < width * num_frames * num_channels:
            raise ValueError("fewer bytes than num_frames * num_channels samples")
        return self._extract(raw, sample_format, num_frames, num_channels, sample_rate, fft_size, fft_overlap,
                             num_coeffs, channels_behavior)

    def _extract(self, a, fmt, frames, channels, sample_rate, fft_size, fft_overlap, num_coeffs, channels_behavior):
        n = max(0, lib.sgz_extract_num_frames(frames, fft_size, fft_overlap))
        out = np.empty((max(n, 1), num_coeffs + 1), dtype=np.float32)
        cnt = C.c_int64(0)
        self._check(lib.sgz_extract_pcm(self._h, N.ptr(a), fmt, frames, channels, sample_rate, fft_size, fft_overlap,
                                        num_coeffs, channels_behavior, N.ptr(out), out.shape[0], C.byref(cnt)))
        return out[: cnt.value]


class Selector:
    """Match bookkeeping of FeatureCorrelationImpl (host side, sgz_sel_*)."""

    def __init__(self, num_matches: int, num_per_file: int, min_spacing: int, step_size: int, target_frames: int):
        self._h = lib.sgz_sel_create(num_matches, num_per_file, min_spacing, step_size, target_frames)
        if not self._h:
            raise ValueError("illegal selector parameters")
        self.num_matches = num_matches

    def close(self) -> None:
        if self._h:
            lib.sgz_sel_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed_dense(self, file: int, sims: np.ndarray, boosts: np.ndarray) -> None:
        s = np.ascontiguousarray(sims, dtype=np.float32)
        b = np.ascontiguousarray(boosts, dtype=np.float32)
        rc = lib.sgz_sel_feed_file(self._h, file, s.shape[0], None, N.ptr(s), N.ptr(b))
        if rc != N.SGZ_OK:
            raise N.NativeError(rc, "sgz_sel_feed_file")

    def feed_candidates(self, file: int, offsets: np.ndarray, sims: np.ndarray, boosts: np.ndarray) -> None:
        o = np.ascontiguousarray(offsets, dtype=np.int32)
        s = np.ascontiguousarray(sims, dtype=np.float32)
        b = np.ascontiguousarray(boosts, dtype=np.float32)
        rc = lib.sgz_sel_feed_file(self._h, file, s.shape[0], N.ptr(o), N.ptr(s), N.ptr(b))
        if rc != N.SGZ_OK:
            raise N.NativeError(rc, "sgz_sel_feed_file")

    def feed_file_punch(self, file: int, num_frames: int, in_frames: int, out_frames: int, min_punch: int,
                        max_punch: int, in_sim: np.ndarray, in_boost: np.ndarray, out_sim: np.ndarray,
                        out_boost: np.ndarray) -> None:
        """One file of a punch-in / punch-out search from the dense results of both targets."""
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (in_sim, in_boost, out_sim, out_boost)]
        rem = num_frames - min_punch  # frames the punch-in pass runs over
        n_in = rem - in_frames + 1 if rem >= in_frames else (1 if rem > 0 else 0)
        if len(a[0]) < n_in or len(a[1]) < n_in or len(a[2]) < max(0, num_frames - out_frames + 1) or \
                len(a[3]) < len(a[2]):
            raise ValueError("dense arrays are shorter than the file's windows")
        rc = lib.sgz_sel_feed_file_punch(self._h, file, num_frames, in_frames, out_frames, min_punch, max_punch,
                                         N.ptr(a[0]), N.ptr(a[1]), N.ptr(a[2]), N.ptr(a[3]))
        if rc != N.SGZ_OK:
            raise N.NativeError(rc, "sgz_sel_feed_file_punch")

    def feed_candidate_list(self, cands: np.ndarray) -> None:
        """All files of a (file, offset)-sorted candidate array in one call."""
        c = np.ascontiguousarray(cands, dtype=N.CANDIDATE_DTYPE)
        rc = lib.sgz_sel_feed_candidates(self._h, N.ptr(c), c.shape[0])
        if rc != N.SGZ_OK:
            raise N.NativeError(rc, "sgz_sel_feed_candidates (candidates not sorted by file, offset?)")

    def feed_screened(self, cands: np.ndarray, head_files: int, num_files: int) -> bool:
        """Feeds the re-scored candidates of Context.corr_compact_bounded.  False: the shortcut does not
        apply, this selector must be discarded and the exact path taken."""
        c = np.ascontiguousarray(cands, dtype=N.CANDIDATE_DTYPE)
        rc = lib.sgz_sel_feed_screened(self._h, N.ptr(c), c.shape[0], head_files, num_files)
        if rc == N.SGZ_E_STATE:
            return False
        if rc != N.SGZ_OK:
            raise N.NativeError(rc, "sgz_sel_feed_screened")
        return True

    def feed_top(self, cands: np.ndarray, threshold: float) -> bool:
        """Feeds the re-scored candidates of Context.corr_compact_top.  False: leaving out the files below the
        threshold was not sound for this data (or a NaN): discard this selector, take corr_compact_bounded."""
        c = np.ascontiguousarray(cands, dtype=N.CANDIDATE_DTYPE)
        rc = lib.sgz_sel_feed_top(self._h, N.ptr(c), c.shape[0], threshold)
        if rc == N.SGZ_E_STATE:
            return False
        if rc != N.SGZ_OK:
            raise N.NativeError(rc, "sgz_sel_feed_top")
        return True

    def feed_gathered_top(self, blocks: np.ndarray, world: int, rows: int, maps: np.ndarray,
                          map_offsets: np.ndarray) -> bool | None:
        """feed_top over the blocks of a candidate exchange (sgz_sel_feed_gathered_top).  None: some rank has
        more records than a block holds; False: the filter does not apply."""
        rc = lib.sgz_sel_feed_gathered_top(self._h, N.ptr(blocks), world, rows, N.ptr(maps), N.ptr(map_offsets))
        if rc == N.SGZ_E_OVERFLOW:
            return None
        if rc == N.SGZ_E_STATE:
            return False
        if rc != N.SGZ_OK:
            raise N.NativeError(rc, "sgz_sel_feed_gathered_top")
        return True

    @property
    def is_full(self) -> bool:
        return bool(lib.sgz_sel_is_full(self._h))

    @property
    def threshold(self) -> float:
        return float(lib.sgz_sel_threshold(self._h))

    def feed_gathered(self, blocks: np.ndarray, world: int, rows: int, maps: np.ndarray, map_offsets: np.ndarray,
                      head_files: int, num_files: int) -> bool | None:
        """feed_screened over the blocks of a candidate exchange (sgz_sel_feed_gathered).  None: some rank has
        more records than a block holds; False: the shortcut does not apply."""
        rc = lib.sgz_sel_feed_gathered(self._h, N.ptr(blocks), world, rows, N.ptr(maps), N.ptr(map_offsets),
                                       head_files, num_files)
        if rc == N.SGZ_E_OVERFLOW:
            return None
        if rc == N.SGZ_E_STATE:
            return False
        if rc != N.SGZ_OK:
            raise N.NativeError(rc, "sgz_sel_feed_gathered")
        return True

    def result(self) -> list[N.Match]:
        arr = (N.Match * max(self.num_matches, 1))()
        n = lib.sgz_sel_result(self._h, arr, self.num_matches)
        return [arr[i] for i in range(n)]


def ring_runs(num_frames: Sequence[int], target_frames: int) -> np.ndarray:
    """Where the ring contents behind the own frames of every file shorter than the target come from, for files
    of the given frame counts in iteration order (sgz_corr_ring_runs): records of N.RING_RUN_DTYPE."""
    fr = np.ascontiguousarray(num_frames, dtype=np.int64)
    n = lib.sgz_corr_ring_runs(N.ptr(fr), len(fr), target_frames, None, 0)
    if n < 0:
        raise N.NativeError(n, "sgz_corr_ring_runs")
    out = np.zeros(max(n, 1), dtype=N.RING_RUN_DTYPE)
    lib.sgz_corr_ring_runs(N.ptr(fr), len(fr), target_frames, N.ptr(out), n)
    return out[:n]


def search_multi(contexts: Sequence[Context], global_ids: Sequence[Sequence[int]], num_matches: int, num_per_file: int,
                 min_spacing: int, step_size: int, capacity: int = 1 << 16) -> list[N.Match]:
    """One search over a database spread over several contexts (one per GPU), from one process
    (sgz_corr_search_multi).  Every context has the same norms and target set; global_ids[c][i] = position of
    context c's file i in the iteration order over all files."""
    n = len(contexts)
    handles = (C.c_void_p * n)(*[c._h for c in contexts])
    arrays = [np.ascontiguousarray(g, dtype=np.int32) for g in global_ids]
    ptrs = (C.c_void_p * n)(*[a.ctypes.data if len(a) else None for a in arrays])
    total = int(sum(len(a) for a in arrays))
    out = (N.Match * max(num_matches, 1))()
    cnt = C.c_int(0)
    rc = lib.sgz_corr_search_multi(handles, n, ptrs, total, contexts[0].target_frames, num_matches, num_per_file,
                                   min_spacing, step_size, capacity, out, num_matches, C.byref(cnt))
    if rc != N.SGZ_OK:
        msgs = "; ".join(lib.sgz_last_error(c._h).decode() for c in contexts)
        raise N.NativeError(rc, msgs)
    return [out[i] for i in range(cnt.value)]


def segm_select(sims: np.ndarray, af_start: int, half_win: int, step_size: int, num_breaks: int,
                min_spacing: int) -> list[N.Break]:
    s = np.ascontiguousarray(sims, dtype=np.float32)
    arr = (N.Break * max(num_breaks, 1))()
    n = lib.sgz_segm_select(N.ptr(s), s.shape[0], af_start, half_win, step_size, num_breaks, min_spacing, arr,
                            num_breaks)
    if n < 0:
        raise N.NativeError(n, "sgz_segm_select")
    return [arr[i] for i in range(n)]


def match_tuple(m: N.Match) -> tuple:
    return (m.sim, m.file, m.start, m.stop, m.boost_in, m.boost_out)


def iter_files(cands: np.ndarray) -> Iterable[tuple[int, np.ndarray]]:
    """Splits a (file, offset)-sorted candidate array into per-file slices."""
    if len(cands) == 0:
        return
    files = cands["file"]
    cuts = np.flatnonzero(np.diff(files)) + 1
    start = 0
    for c in list(cuts) + [len(cands)]:
        yield int(files[start]), cands[start:c]
        start = c
