"""Second, independent CPU restatement of the reference's similarity path, in numpy / plain Python.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/ to cross-check
strugatzki_oracle.c.  It was written from the reference's Scala sources, not from the C file, so that a
slip in one transcription shows up as a disagreement between the two.  It cannot pin either of them to the
reference (parity unpinned: the reference cannot run here and holds no numeric fixtures); it is the only
second witness available.

Paths are relative to src/main/scala/de/sciss/strugatzki/ of the reference.  Matrices are `mat[ch][frame]`
float32 arrays, as in the reference; the drivers take feature files as [frames][ch] arrays and transpose.

Order of operations: every reference loop adds into one `var sum = 0.0` sequentially.  numpy's `sum` adds
pairwise, so the sequential order is reproduced with `cumsum` (which is sequential) over float64 values.
"""
from __future__ import annotations

import math
import struct

import numpy as np

F32 = np.float32


def _seq_sum(x: np.ndarray) -> float:
    """x[0] + x[1] + ... in that order, fp64 (0.0 + x[0] == x[0])."""
    x = np.asarray(x, dtype=np.float64).ravel()
    return float(np.cumsum(x)[-1]) if x.size else 0.0


# ---- impl/MathUtil.scala -------------------------------------------------------------------------------------
def stat(mat: np.ndarray, frame_off: int, frame_len: int, chan_off: int, chan_len: int) -> tuple[float, float]:
    """MathUtil.stat, impl/MathUtil.scala:29-62: one mean and one population std-dev over the whole
    chan_len x frame_len sub-matrix, rows outermost; Float cells widen to Double in `sum += cb(i)` and in
    `cb(i) - mean`."""
    sub = np.asarray(mat, dtype=F32)[chan_off:chan_off + chan_len, frame_off:frame_off + frame_len].astype(np.float64)
    size = frame_len * chan_len  # Int
    mean = _seq_sum(sub) / size if size else float("nan")  # 0.0 / 0 on the JVM: NaN
    d = sub - mean
    var = _seq_sum(d * d) / size if size else float("nan")
    return mean, math.sqrt(var) if var == var and var >= 0 else float("nan")


def correlate(a: np.ndarray, a_mean: float, a_std: float, num_frames: int, num_channels: int, b: np.ndarray,
              b_mean: float, b_std: float, b_frame_off: int, b_chan_off: int) -> np.float32:
    """MathUtil.correlate, impl/MathUtil.scala:177-196: sum over rows, then frames, of
    (a(ch)(i) - aMean) * (b(ch + bChanOff)((i + bFrameOff) % b.length) - bMean), divided by
    aStdDev * bStdDev * (numChannels * numFrames), `.toFloat`."""
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    blen = b.shape[1]
    idx = (np.arange(num_frames) + b_frame_off) % blen
    ca = a[:num_channels, :num_frames].astype(np.float64) + (-a_mean)
    cb = b[b_chan_off:b_chan_off + num_channels][:, idx].astype(np.float64) + (-b_mean)
    s = _seq_sum(ca * cb)
    with np.errstate(divide="ignore", invalid="ignore"):
        return F32(np.float64(s) / (np.float64(a_std) * np.float64(b_std) * (num_channels * num_frames)))


def correlate_half(num_channels: int, half: int, a: np.ndarray, frame_off: int, chan_off: int) -> np.float32:
    """MathUtil.correlateHalf, impl/MathUtil.scala:80-99: joint mean / std-dev over the 2 * half frames
    (stat over columns 0 .. 2 half - 1, whatever frameOff is), cross term between the half starting at
    frameOff and the half starting half frames later, both indices modulo 2 * half; divided by
    stdDev * stdDev * (numChannels * half)."""
    a = np.asarray(a, dtype=F32)
    n = half << 1
    mean, sd = stat(a, 0, n, chan_off, num_channels)
    i = np.arange(frame_off, frame_off + half)
    rows = a[chan_off:chan_off + num_channels].astype(np.float64)
    x = rows[:, i % n] + (-mean)
    y = rows[:, (i + half) % n] + (-mean)
    s = _seq_sum(x * y)
    with np.errstate(divide="ignore", invalid="ignore"):
        return F32(np.float64(s) / (np.float64(sd) * np.float64(sd) * (num_channels * half)))


def avg(b: np.ndarray, off: int, length: int) -> np.float32:
    """MathUtil.avg, impl/MathUtil.scala:109-118: fp64 sum / len, `.toFloat`."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return F32(np.float64(_seq_sum(np.asarray(b, dtype=F32)[off:off + length])) / np.float64(length))


def normalize(norm_buf: np.ndarray | None, b: np.ndarray, b_off: int, b_len: int) -> None:
    """MathUtil.normalize, impl/MathUtil.scala:132-152, in place: (f - min) / (max - min), all Float."""
    if norm_buf is None:
        return
    nb = np.asarray(norm_buf, dtype=F32)
    for ch in range(b.shape[0]):
        mn, mx = nb[ch][0], nb[ch][1]
        d = F32(mx - mn)
        with np.errstate(divide="ignore", invalid="ignore"):
            b[ch, b_off:b_off + b_len] = (b[ch, b_off:b_off + b_len] - mn) / d


def full_to_feat(n: int, step: int) -> int:
    """impl/FeatureCorrelationImpl.scala:38: ((n + (stepSize >> 1)) / stepSize).toInt, Long division
    (truncating towards zero)."""
    v = n + (step >> 1)
    q = abs(v) // step
    return q if v >= 0 else -q


def feat_to_full(i: int, step: int) -> int:
    return i * step


# ---- the sorted sets ------------------------------------------------------------------------------------------
def float_compare(a, b) -> int:
    """java.lang.Float.compare: numeric order, then -0.0 < 0.0, NaN equal to itself and above everything."""
    a, b = float(F32(a)), float(F32(b))
    if a < b:
        return -1
    if a > b:
        return 1

    def bits(x):
        if x != x:
            return 0x7fc00000  # floatToIntBits canonicalises NaN
        (u,) = struct.unpack("<i", struct.pack("<f", x))
        return u

    ia, ib = bits(a), bits(b)
    return 0 if ia == ib else (-1 if ia < ib else 1)


class SimSet:
    """immutable.SortedSet[Match] / [Break] under an Ordering that looks at `sim` only
    (MatchMinOrd: b.sim compare a.sim, FeatureCorrelation.scala:75-77 -- best = highest first;
    BreakMaxOrd: a.sim compare b.sim, FeatureSegmentation.scala:60-62 -- best = lowest first).
    Two elements with equal sims are the same key: `+` leaves the set as it is (the tree keeps the key it
    has), `-` removes the element that compares equal."""

    def __init__(self, highest_first: bool):
        self.highest_first = highest_first
        self.items: list = []  # kept in set order

    def _cmp(self, x, y) -> int:
        return float_compare(y.sim, x.sim) if self.highest_first else float_compare(x.sim, y.sim)

    def add(self, e) -> None:
        for k, have in enumerate(self.items):
            c = self._cmp(have, e)
            if c == 0:
                return
            if c > 0:
                self.items.insert(k, e)
                return
        self.items.append(e)

    def remove(self, e) -> None:
        for k, have in enumerate(self.items):
            if self._cmp(have, e) == 0:
                del self.items[k]
                return

    def add_all(self, other: "SimSet") -> None:
        for e in other.items:
            self.add(e)

    def take(self, n: int) -> None:
        del self.items[n:]

    @property
    def last(self):
        return self.items[-1]

    def __len__(self):
        return len(self.items)


class Match:
    __slots__ = ("sim", "file", "start", "stop", "boost_in", "boost_out")

    def __init__(self, sim, file, start, stop, boost_in, boost_out):
        self.sim, self.file, self.start, self.stop = F32(sim), file, start, stop
        self.boost_in, self.boost_out = F32(boost_in), F32(boost_out)

    def tuple(self):
        return (float(self.sim), self.file, self.start, self.stop, float(self.boost_in), float(self.boost_out))


class Break:
    __slots__ = ("sim", "pos")

    def __init__(self, sim, pos):
        self.sim, self.pos = F32(sim), pos


def spacing(a_start, a_stop, b_start, b_stop) -> int:
    """SpanUtil.spacing, impl/SpanUtil.scala:38-43: from the stop of the span that starts first to the start
    of the other (negative when they overlap)."""
    if a_start < b_start:
        return b_start - a_stop
    return a_start - b_stop


# ---- impl/FeatureCorrelationImpl.scala (no punch-out) ------------------------------------------------------------
def correlation_run(target: np.ndarray, punch_start: int, punch_stop: int, weight: float, norms, db,
                    max_boost: float, num_matches: int, num_per_file: int, min_spacing: int, step: int):
    """body() of impl/FeatureCorrelationImpl.scala:32-412 for punchOut = None.  `target`: feature file of the
    input ([frames][ch]); `db`: list of feature files in iteration order.  Returns (matches, sims, boosts)
    with one dense array pair per file, one value per loop iteration."""
    w = F32(weight)
    mb = F32(max_boost)
    norm_buf = None if norms is None else np.asarray(norms, dtype=F32)

    # readInBuffer, :83-98
    start = full_to_feat(punch_start, step)
    stop = full_to_feat(punch_stop, step)
    L = stop - start
    tb = np.array(np.asarray(target, dtype=F32)[start:stop].T, dtype=F32, order="C")
    normalize(norm_buf, tb, 0, L)
    t_mean, t_std = stat(tb[:1], 0, L, 0, 1)
    s_mean, s_std = stat(tb[1:], 0, L, 0, tb.shape[0] - 1)
    ln_avg_in = _jlog(float(avg(tb[0], 0, L)))  # calcLnAvgLoud, :73
    num_ch = tb.shape[0]

    all_prio = SimSet(True)
    entry = SimSet(True)
    ring = np.zeros((num_ch, L), dtype=F32)  # eInBuf, :154 -- allocated once, shared by all files
    sims_out, boosts_out = [], []

    for file_idx, feat in enumerate(db):
        feat = np.asarray(feat, dtype=F32)
        entry = SimSet(True)  # :166
        last = None           # :167
        left = feat.shape[0]  # :182
        read_sz, read_off, logical, pos = L, 0, 0, 0
        fs, fb = [], []
        while left > 0:  # :190
            chunk = min(left, read_sz)
            ring[:, read_off:read_off + chunk] = feat[pos:pos + chunk].T  # afExtr.read, :195
            pos += chunk
            b_off = logical % L
            normalize(norm_buf, ring, read_off, chunk)  # :197
            # calcBoost, :73-78
            ln_avg_b = _jlog(float(avg(ring[0], 0, L)))
            boost = _to_f32(_jexp((ln_avg_in - ln_avg_b) / 0.6))
            if boost <= mb:  # :199 (false for NaN)
                temporal = F32(0)
                spectral = F32(0)
                if w > F32(0):
                    bm, bs = stat(ring, 0, L, 0, 1)  # :419 -- frame offset 0 whatever the rotation
                    temporal = correlate(tb[:1], t_mean, t_std, L, 1, ring, bm, bs, b_off, 0)
                if w < F32(1):
                    bm, bs = stat(ring, 0, L, 1, num_ch - 1)
                    spectral = correlate(tb[1:], s_mean, s_std, L, num_ch - 1, ring, bm, bs, b_off, 1)
                with np.errstate(invalid="ignore", over="ignore"):
                    sim = F32(F32(temporal * w) + F32(spectral * F32(F32(1) - w)))  # :207, Float arithmetic
            else:
                sim = F32(0)  # :209
            fs.append(sim)
            fb.append(boost)
            # :234-239
            max_entry = min(num_matches - len(all_prio), num_per_file)
            has_space = len(entry) < max_entry
            lowest = entry.last.sim if len(entry) else (all_prio.last.sim if len(all_prio) else F32(0))
            if has_space or sim > lowest:
                m = Match(sim, file_idx, feat_to_full(logical, step), feat_to_full(logical + L, step), boost, 1.0)
                # addMatch, :135-150
                if last is not None and spacing(m.start, m.stop, last.start, last.stop) < min_spacing:
                    if last.sim < m.sim:
                        entry.remove(last)
                        entry.add(m)
                        last = m
                else:
                    entry.add(m)
                    if len(entry) > num_per_file:
                        entry.remove(entry.last)
                    last = m
            left -= chunk
            read_off = (read_off + chunk) % L
            logical += 1
            read_sz = 1
        all_prio.add_all(entry)  # :399
        if len(all_prio) > num_matches:
            all_prio.take(num_matches)  # :400
        sims_out.append(np.array(fs, dtype=F32))
        boosts_out.append(np.array(fb, dtype=F32))
    return list(all_prio.items), sims_out, boosts_out


def _jlog(x: float) -> float:
    """math.log on the JVM: NaN for negative arguments and NaN, -Infinity for 0."""
    if x != x or x < 0:
        return float("nan")
    if x == 0:
        return float("-inf")
    return math.log(x)


def _jexp(x: float) -> float:
    if x != x:
        return float("nan")
    try:
        return math.exp(x)
    except OverflowError:
        return float("inf")


def _to_f32(x: float) -> np.float32:
    with np.errstate(over="ignore", invalid="ignore"):
        return F32(x)


# ---- impl/FeatureSegmentationImpl.scala ---------------------------------------------------------------------
def segmentation_run(feat: np.ndarray, span_start, span_stop, corr_len: int, weight: float, norms,
                     num_breaks: int, min_spacing: int, step: int):
    """body() of impl/FeatureSegmentationImpl.scala:31-142; span_start / span_stop None = open.
    Returns (breaks, dense sims)."""
    feat = np.asarray(feat, dtype=F32)
    w = F32(weight)
    norm_buf = None if norms is None else np.asarray(norms, dtype=F32)
    half = full_to_feat(corr_len, step)
    win = half * 2
    num_ch = feat.shape[1]
    ring = np.zeros((num_ch, win), dtype=F32)  # :86
    af_start = 0 if span_start is None else max(0, full_to_feat(span_start, step))          # :90-93
    af_stop = feat.shape[0] if span_stop is None else min(feat.shape[0], full_to_feat(span_stop, step))  # :94-97
    left = af_stop - af_start
    prio = SimSet(False)
    last = None
    read_sz, read_off, logical, pos = win, 0, 0, af_start
    dense = []
    while left > 0:  # :107
        chunk = min(left, read_sz)
        ring[:, read_off:read_off + chunk] = feat[pos:pos + chunk].T
        pos += chunk
        off = logical % win
        normalize(norm_buf, ring, read_off, chunk)
        temporal = correlate_half(1, half, ring, off, 0) if w > F32(0) else F32(0)
        spectral = correlate_half(num_ch - 1, half, ring, off, 1) if w < F32(1) else F32(0)
        with np.errstate(invalid="ignore", over="ignore"):
            sim = F32(F32(temporal * w) + F32(spectral * F32(F32(1) - w)))  # :120
        dense.append(sim)
        has_space = len(prio) < num_breaks                       # :58
        highest = prio.last.sim if len(prio) else F32(0)         # :60-62
        if has_space or sim < highest:                           # :121
            b = Break(sim, feat_to_full(af_start + logical + half, step))
            if last is not None and (b.pos - last.pos) < min_spacing:  # :69
                if last.sim > b.sim:
                    prio.remove(last)
                    prio.add(b)
                    last = b
            else:
                prio.add(b)
                if len(prio) > num_breaks:
                    prio.remove(prio.last)
                last = b
        left -= chunk
        read_off = (read_off + chunk) % win
        logical += 1
        read_sz = 1
    return list(prio.items), np.array(dense, dtype=F32)


# ---- impl/SelfSimilarityImpl.scala -----------------------------------------------------------------------------
def selfsim_geometry(n1: int, n2: int, span_start, span_stop, corr_len: int, decimation: int, step: int):
    """:64-91 -> (img_ext, decim, af_start, num_corrs, half)."""
    half = full_to_feat(corr_len, step)
    win = half * 2
    frames = min(n1, n2)
    af_start = 0 if span_start is None else max(0, full_to_feat(span_start, step))
    af_stop = frames if span_stop is None else min(frames, full_to_feat(span_stop, step))
    num_corrs = max(0, (af_stop - af_start) - win + 1)
    d = decimation
    i = num_corrs // d
    if i > 0xB504:
        d = (num_corrs + 0xB503) // 0xB504
        i = num_corrs // d
    return i, d, af_start, num_corrs, half


def gray_pixel(sim, color_warp: float, color_ceil: float, inverted: bool) -> int:
    """:98-107, :114-115, :150.  math.pow(max(0f, sim), colorWarp).toFloat * colorScale is Float; `v * 255` is
    Float * Int = Float, `+ 0.5` widens to Double; `.toInt` truncates (NaN -> 0, saturating)."""
    scale = F32(F32(1.0) / F32(color_ceil))
    m = float(max(F32(0), F32(sim))) if sim == sim else float("nan")  # math.max(0f, NaN) = NaN
    with np.errstate(over="ignore", invalid="ignore"):
        v = F32(F32(_jpow(m, float(F32(color_warp)))) * scale)
        if inverted:
            v = F32(F32(1) - v)
        d = float(F32(v * F32(255))) + 0.5
    if d != d:
        i = 0
    elif d >= 2147483647.0:
        i = 2147483647
    elif d <= -2147483648.0:
        i = -2147483648
    else:
        i = int(d)  # truncation towards zero
    i = max(0, min(255, i))
    return (i << 16) | (i << 8) | i


def _jpow(x: float, y: float) -> float:
    if x != x:
        return float("nan") if y != 0 else 1.0
    try:
        return math.pow(x, y)
    except (OverflowError, ValueError):
        return float("inf")


def selfsim_run(feat1: np.ndarray, feat2: np.ndarray, span_start, span_stop, corr_len: int, decimation: int,
                weight: float, norms, color_warp: float, color_ceil: float, color_inv: bool, step: int):
    """body() of impl/SelfSimilarityImpl.scala:31-180 with the gray colour scheme.
    Returns (sims [x][y] float32 with x = left / decim, raster int32 [row][col] as the reference fills it)."""
    f1 = np.asarray(feat1, dtype=F32)
    f2 = np.asarray(feat2, dtype=F32)
    w = F32(weight)
    norm_buf = None if norms is None else np.asarray(norms, dtype=F32)
    ext, decim, af_start, num_corrs, half = selfsim_geometry(f1.shape[0], f2.shape[0], span_start, span_stop,
                                                            corr_len, decimation, step)
    num_ch = f1.shape[1]
    ring = np.zeros((num_ch, 2 * half), dtype=F32)
    sims = np.zeros((ext, ext), dtype=F32)
    img = np.zeros((ext, ext), dtype=np.int32)
    stop = num_corrs // decim * decim
    for left in range(0, stop, decim):
        ring[:, :half] = f1[left + af_start:left + af_start + half].T
        normalize(norm_buf, ring, 0, half)
        for right in range(left, stop, decim):
            ring[:, half:] = f2[right + af_start:right + af_start + half].T
            normalize(norm_buf, ring, half, half)
            temporal = correlate_half(1, half, ring, 0, 0) if w > F32(0) else F32(0)
            spectral = correlate_half(num_ch - 1, half, ring, 0, 1) if w < F32(1) else F32(0)
            with np.errstate(invalid="ignore", over="ignore"):
                sim = F32(F32(temporal * w) + F32(spectral * F32(F32(1) - w)))
            colr = gray_pixel(sim, color_warp, color_ceil, color_inv)
            x, y = left // decim, right // decim
            sims[x, y] = sim
            sims[y, x] = sim
            img[ext - 1 - y, x] = colr  # off1 = (imgExtM1 - rightOff / decim) * imgExt + leftOff / decim
            img[ext - 1 - x, y] = colr  # off2
    return sims, img
