"""The numpy / Python twin (oracle/twin.py, written from the Scala sources) against the C oracle
(oracle/strugatzki_oracle.c) and against values derived by hand from the reference's formulas.

Neither restatement can be pinned to the reference here (it cannot run, and holds no numeric fixtures); two
independent transcriptions that agree bit for bit on the committed vectors, on random cases and on hand-made
corner cases (equal similarities, NaN, -0.0, overlapping spans) are the strongest check available.
CPU only.
"""
import math
import os

import numpy as np
import pytest

from common import NUM_CH, SEED, STEP, O, corr_config, make_plants, norms_default, synth_db, synth_target
from oracle import twin as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32 = np.float32


# ---- hand-computed cases -------------------------------------------------------------------------------------
def test_stat_by_hand():
    # impl/MathUtil.scala:29-62 over [[1, 2, 3, 4]]: sum 10, mean 10 / 4 = 2.5; deviations -1.5 -0.5 0.5 1.5,
    # squares 2.25 + 0.25 + 0.25 + 2.25 = 5, std-dev sqrt(5 / 4) = sqrt(1.25)
    mat = np.array([[1, 2, 3, 4]], dtype=F32)
    for fn in (T.stat, O.stat):
        m, s = fn(mat, 0, 4, 0, 1)
        assert m == 2.5 and s == math.sqrt(1.25)
    # two rows, one statistic over all 6 cells (not per row): [[1, 2, 3], [5, 7, 9]] sum 27, mean 4.5;
    # squares 12.25 + 6.25 + 2.25 + 0.25 + 6.25 + 20.25 = 47.5, std-dev sqrt(47.5 / 6)
    mat = np.array([[1, 2, 3], [5, 7, 9]], dtype=F32)
    for fn in (T.stat, O.stat):
        m, s = fn(mat, 0, 3, 0, 2)
        assert m == 4.5 and s == math.sqrt(47.5 / 6)
    # sub-matrix: rows 1..1, frames 1..2 -> [7, 9]: mean 8, std-dev 1
    for fn in (T.stat, O.stat):
        assert fn(mat, 1, 2, 1, 1) == (8.0, 1.0)


def test_correlate_by_hand():
    # impl/MathUtil.scala:177-196 with a = [1, 2, 3] (mean 2, std sqrt(2/3)) and b = [2, 4, 9] (mean 5,
    # squares 9 + 1 + 16 = 26, std sqrt(26/3)), no rotation:
    # sum = (-1)(-3) + (0)(-1) + (1)(4) = 7;  7 / (sqrt(2/3) * sqrt(26/3) * 3) = 7 / sqrt(52) = 0.97072...
    a = np.array([[1, 2, 3]], dtype=F32)
    b = np.array([[2, 4, 9]], dtype=F32)
    am, asd = T.stat(a, 0, 3, 0, 1)
    bm, bsd = T.stat(b, 0, 3, 0, 1)
    want = F32(7.0 / (math.sqrt(2 / 3) * math.sqrt(26 / 3) * 3))
    assert abs(float(want) - 7 / math.sqrt(52)) < 1e-7
    assert T.correlate(a, am, asd, 3, 1, b, bm, bsd, 0, 0) == want
    assert O.correlate(a, am, asd, b, bm, bsd, 0, 0) == want
    # the ring rotated by one: b is read as [4, 9, 2] -> sum = (-1)(-1) + 0 + (1)(-3) = -2
    want = F32(-2.0 / (math.sqrt(2 / 3) * math.sqrt(26 / 3) * 3))
    assert T.correlate(a, am, asd, 3, 1, b, bm, bsd, 1, 0) == want
    assert O.correlate(a, am, asd, b, bm, bsd, 1, 0) == want
    # correlate(a, a) == 1 (the comment at :195) and correlate(a, -a) == -1
    assert T.correlate(a, am, asd, 3, 1, a, am, asd, 0, 0) == F32(1)
    assert T.correlate(a, am, asd, 3, 1, -a, -am, asd, 0, 0) == F32(-1)


def test_correlate_half_by_hand():
    # impl/MathUtil.scala:80-99, one row [1, 2, 4, 5], half = 2: joint mean 3, squares 4 + 1 + 1 + 4 = 10,
    # variance 2.5; cross = (1-3)(4-3) + (2-3)(5-3) = -2 - 2 = -4; -4 / (2.5 * 2) = -0.8
    a = np.array([[1, 2, 4, 5]], dtype=F32)
    assert T.correlate_half(1, 2, a, 0, 0) == F32(-0.8)
    assert O.correlate_half(1, 2, a, 0, 0) == F32(-0.8)
    # frameOff 1 rotates both halves: first half = columns 1, 2, second = columns 3, 0:
    # (2-3)(5-3) + (4-3)(1-3) = -2 - 2 = -4 -> the same -0.8
    assert T.correlate_half(1, 2, a, 1, 0) == F32(-0.8)
    assert O.correlate_half(1, 2, a, 1, 0) == F32(-0.8)
    # identical halves do not give 1 unless each half has the joint mean: [1, 3, 1, 3] -> mean 2, var 1,
    # cross = (-1)(-1) + (1)(1) = 2, 2 / (1 * 2) = 1
    a = np.array([[1, 3, 1, 3]], dtype=F32)
    assert T.correlate_half(1, 2, a, 0, 0) == F32(1) == O.correlate_half(1, 2, a, 0, 0)


def test_avg_normalize_full_to_feat_by_hand():
    b = np.array([0.5, 1.5, 4.0, 100.0], dtype=F32)
    assert T.avg(b, 0, 3) == F32(2.0) == O.avg(b, 0, 3)
    # impl/MathUtil.scala:132-152: (f - min) / (max - min) in Float, no clipping
    mat = np.array([[0.0, 1.5, 3.0], [10.0, 20.0, 40.0]], dtype=F32)
    norms = np.array([[0.0, 1.5], [10.0, 30.0]], dtype=F32)
    got = mat.copy()
    T.normalize(norms, got, 1, 2)  # frames 1, 2 only
    assert np.array_equal(got, np.array([[0.0, 1.0, 2.0], [10.0, 0.5, 1.5]], dtype=F32))
    assert np.array_equal(O.normalize(norms, mat, 1, 2), got)
    # impl/FeatureCorrelationImpl.scala:38 with stepSize 512: (n + 256) / 512
    for n, want in ((0, 0), (255, 0), (256, 1), (767, 1), (768, 2), (44100, 86), (22050, 43), (1323000, 2584)):
        assert T.full_to_feat(n, STEP) == want == O.full_to_feat(n, STEP)


def test_float_compare_total_order():
    nan, inf = float("nan"), float("inf")
    order = [-inf, -1.0, -0.0, 0.0, 1e-45, 1.0, inf, nan]  # ascending under java.lang.Float.compare
    for i, x in enumerate(order):
        for j, y in enumerate(order):
            assert T.float_compare(x, y) == (i > j) - (i < j)


class _E:
    def __init__(self, sim, tag):
        self.sim, self.tag = F32(sim), tag


def test_sorted_set_keeps_the_existing_element_on_equal_sims():
    s = T.SimSet(True)
    for sim, tag in ((0.5, "a"), (0.7, "b"), (0.5, "c"), (float("nan"), "n"), (0.0, "z"), (-0.0, "m")):
        s.add(_E(sim, tag))
    # NaN is the largest float under Float.compare -> first in a highest-first set; 0.5 "c" was dropped;
    # -0.0 < 0.0 are different keys
    assert [e.tag for e in s.items] == ["n", "b", "a", "z", "m"]
    s.remove(_E(0.5, "whatever"))  # removal is by key
    assert [e.tag for e in s.items] == ["n", "b", "z", "m"]
    lo = T.SimSet(False)
    for sim, tag in ((0.5, "a"), (0.2, "b"), (0.2, "c")):
        lo.add(_E(sim, tag))
    assert [e.tag for e in lo.items] == ["b", "a"]


# ---- the drivers, twin against C oracle -------------------------------------------------------------------------
def _twin_corr(cfg, tgt, norms, db):
    return T.correlation_run(tgt, cfg.punch_in_start, cfg.punch_in_stop, cfg.temporal_weight,
                             norms if cfg.normalize else None, db, cfg.max_boost, cfg.num_matches, cfg.num_per_file,
                             cfg.min_spacing, cfg.step_size)


def _same_matches(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        gt, wt = g.tuple(), O.match_tuple(w)
        assert gt[1:4] == wt[1:4]
        for x, y in ((gt[0], wt[0]), (gt[4], wt[4]), (gt[5], wt[5])):
            assert x == y or (x != x and y != y)


def test_twin_reproduces_the_golden_correlation_vectors():
    g = np.load(os.path.join(GOLD, "correlation.npz"))
    db = [x for x in g["db"]]
    L, ts = int(g["L"]), int(g["ts"])
    for k in range(3):
        nm, npf, ms = (int(v) for v in g[f"setting{k}"])
        cfg = corr_config(ts, L, num_matches=nm, num_per_file=npf, min_spacing=ms)
        m, sims, boosts = _twin_corr(cfg, g["target"], g["norms"], db)
        assert np.array_equal(np.stack(sims), g["sims"]) and np.array_equal(np.stack(boosts), g["boosts"])
        assert np.array_equal(np.array([x.tuple() for x in m], dtype=np.float64), g[f"matches{k}"])
    cfg = corr_config(ts, L, weight=0.2, normalize=False, num_matches=3)
    _, sims, _ = _twin_corr(cfg, g["target"], None, db)
    assert np.array_equal(np.stack(sims), g["sims_raw_w02"])


def test_twin_reproduces_the_golden_segmentation_and_selfsim_vectors():
    g = np.load(os.path.join(GOLD, "segmentation.npz"))
    a0, a1, half, nb, ms = (int(v) for v in g["params"])
    breaks, dense = T.segmentation_run(g["frames"], a0 * STEP, a1 * STEP, 22050, 0.5, g["norms"], nb, ms, STEP)
    assert np.array_equal(dense, g["dense"])
    assert np.array_equal(np.array([(float(b.sim), b.pos) for b in breaks], dtype=np.float64), g["breaks"])
    g = np.load(os.path.join(GOLD, "selfsim.npz"))
    cfg = O.SelfSimConfig(0, 0, 0, 0, 8 * STEP, 3, 0.5, 0, 0.7, 0.9, 0, 1, STEP, NUM_CH - 1)
    want_sim, want_img = O.selfsim_run(cfg, g["frames"], None, g["norms"])
    sim, img = T.selfsim_run(g["frames"], g["frames"], None, None, 8 * STEP, 3, 0.5, g["norms"], 0.7, 0.9, False, STEP)
    assert np.array_equal(sim, want_sim) and np.array_equal(img, want_img)
    assert np.array_equal(sim, g["sim"]) and np.array_equal(img, g["img"])


@pytest.mark.parametrize("seed", range(6))
def test_twin_equals_c_oracle_on_random_searches(seed):
    """Random geometry incl. files shorter than the target (stale ring), empty files, weights 0 / 1, no
    normalisation, a tight boost gate (many exact 0f similarities: equal keys in the sorted sets), minSpacing
    above and below the window overlap."""
    rng = np.random.default_rng(1000 + seed)
    L = int(rng.choice([1, 2, 7, 24]))
    nf = int(rng.integers(1, 7))
    frames = [int(rng.integers(0, 4 * L + 60)) for _ in range(nf)]
    ts = int(rng.integers(0, 9))
    plants = [q for q in make_plants(int(rng.integers(0, 3)), L, nf, max(frames + [L + 1]), ts)
              if q["file"] < nf and q["offset"] + L <= frames[q["file"]]]
    db = synth_db(nf, frames, plants, seed=int(rng.integers(1, 1 << 30)))
    if seed % 3 == 2 and frames[0] > 12:
        db[0][3:12] = db[0][3]  # a constant stretch: zero variance, NaN similarities
    tgt = synth_target(ts, L)
    cfg = corr_config(ts, L, weight=float(rng.choice([0.0, 0.5, 1.0, 0.3])), normalize=bool(rng.integers(0, 2)),
                      max_boost=float(rng.choice([8.0, 1.02])), num_matches=int(rng.integers(1, 7)),
                      num_per_file=int(rng.integers(1, 4)), min_spacing=int(rng.choice([0, 3 * STEP, -2 * STEP])))
    norms = norms_default()
    want, ws, wb = O.correlation_run(cfg, tgt, norms if cfg.normalize else None, db)
    got, gs, gb = _twin_corr(cfg, tgt, norms, db)
    for f in range(nf):
        assert np.array_equal(gs[f], ws[f], equal_nan=True), f
        assert np.array_equal(gb[f], wb[f], equal_nan=True), f
    _same_matches(got, want)


def test_twin_equals_c_oracle_on_equal_and_nan_similarities():
    """Duplicate files (equal best similarities collapse to one set entry: the later file's match is dropped),
    a file that starts with a constant stretch while the set has room (its NaN match is kept, and ranks first)."""
    L, ts = 12, 3
    db = synth_db(4, 90, make_plants(1, L, 4, 90, ts))
    db[1] = db[0].copy()
    db[2][0:40] = db[2][0]
    tgt = synth_target(ts, L)
    norms = norms_default()
    for nm, npf in ((3, 1), (4, 2), (6, 3)):
        cfg = corr_config(ts, L, num_matches=nm, num_per_file=npf)
        want, _, _ = O.correlation_run(cfg, tgt, norms, db)
        got, _, _ = _twin_corr(cfg, tgt, norms, db)
        _same_matches(got, want)
        assert all(m.file != 1 or any(abs(m.sim - o.sim) > 0 for o in got if o.file == 0) for m in got)
    assert any(m.sim != m.sim for m in got)


@pytest.mark.parametrize("seed", range(4))
def test_twin_equals_c_oracle_on_random_segmentations_and_selfsims(seed):
    rng = np.random.default_rng(50 + seed)
    n = int(rng.integers(20, 140))
    frames = O.synth_file(SEED + seed, 1, n, NUM_CH)
    if seed == 1:
        frames[10:30] = frames[10]
    norms = norms_default() if seed % 2 == 0 else None
    corr_len = int(rng.choice([2, 5, 9])) * STEP
    weight = float(rng.choice([0.0, 0.5, 1.0]))
    a0, a1 = int(rng.integers(0, 8)), n - int(rng.integers(0, 8))
    nb, ms = int(rng.integers(1, 6)), int(rng.choice([0, 4 * STEP, 30 * STEP]))
    cfg = O.SegmConfig(1, 1, a0 * STEP, a1 * STEP, corr_len, weight, int(norms is not None), nb, ms, STEP, NUM_CH - 1)
    want, wd = O.segmentation_run(cfg, frames, norms)
    got, gd = T.segmentation_run(frames, a0 * STEP, a1 * STEP, corr_len, weight, norms, nb, ms, STEP)
    assert np.array_equal(gd, wd, equal_nan=True)
    assert [(float(b.sim), b.pos) for b in got] == [(b.sim, b.pos) for b in want] or \
        all((g.pos == w.pos) and (g.sim == w.sim or (g.sim != g.sim and w.sim != w.sim)) for g, w in zip(got, want))
    decim = int(rng.integers(1, 4))
    warp, ceil, inv = float(rng.choice([1.0, 0.5, 2.0])), float(rng.choice([1.0, 0.8])), bool(rng.integers(0, 2))
    sc = O.SelfSimConfig(0, 0, 0, 0, corr_len, decim, weight, 0, warp, ceil, int(inv), int(norms is not None), STEP,
                         NUM_CH - 1)
    ws, wi = O.selfsim_run(sc, frames, None, norms)
    gs, gi = T.selfsim_run(frames, frames, None, None, corr_len, decim, weight, norms, warp, ceil, inv, STEP)
    assert np.array_equal(gs, ws, equal_nan=True) and np.array_equal(gi, wi)


def test_gray_pixel_by_hand():
    # impl/SelfSimilarityImpl.scala:99-107,150: v = pow(max(0f, sim), warp).toFloat * (1f / ceil);
    # (v * 255 + 0.5).toInt truncates: 0.5 -> 127.5 + 0.5 = 128; 0.498 -> 126.99 + 0.5 = 127.49 -> 127
    assert T.gray_pixel(F32(0.5), 1.0, 1.0, False) == 0x808080
    assert T.gray_pixel(F32(0.498), 1.0, 1.0, False) == 0x7f7f7f
    assert T.gray_pixel(F32(-0.3), 1.0, 1.0, False) == 0           # max(0f, sim)
    assert T.gray_pixel(F32(1.7), 1.0, 1.0, False) == 0xffffff      # clamped to 255
    assert T.gray_pixel(F32(0.25), 0.5, 1.0, False) == 0x808080     # sqrt(0.25) = 0.5
    assert T.gray_pixel(F32(0.25), 1.0, 0.5, False) == 0x808080     # 0.25 / 0.5
    assert T.gray_pixel(F32(0.25), 1.0, 1.0, True) == 0xbfbfbf      # 1 - 0.25 = 0.75 -> 191.25 + 0.5 -> 191
    assert T.gray_pixel(F32("nan"), 1.0, 1.0, False) == 0           # NaN.toInt = 0
