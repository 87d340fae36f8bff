"""CPU tests of the oracle against the invariants the reference states itself.

The reference's own suite holds no numeric vectors (StrugatzkiSuite.scala checks
XML round trips only), so these known answers are what pins the restatement:
  * correlate(a, a) == 1.0            impl/MathUtil.scala:98,195
  * boost = exp((ln l_in - ln l_db) / 0.6): a copy whose loudness is l * d**0.6
    gets boost 1/d                    notes/BoostFactor.txt:34-39
  * normalize is (f - min) / (max - min) in fp32, no clipping   impl/MathUtil.scala:146-147
  * matches come back in descending, breaks in ascending similarity
"""
import numpy as np
import pytest

from common import NUM_CH, SEED, STEP, O, corr_config, make_plants, norms_default, synth_db, synth_target


def _rand(shape, seed=0):
    return np.random.default_rng(seed).random(shape, dtype=np.float32)


def test_stat_matches_numpy_two_pass():
    m = _rand((5, 40))
    mean, std = O.stat(m, 3, 30, 1, 3)
    sub = m[1:4, 3:33].astype(np.float64)
    assert mean == pytest.approx(sub.mean(), rel=1e-14)
    assert std == pytest.approx(sub.std(), rel=1e-13)


def test_correlate_identity_is_one():
    a = _rand((13, 64), 1)
    mean, std = O.stat(a, 0, 64, 0, 13)
    assert O.correlate(a, mean, std, a, mean, std, 0, 0) == 1.0


def test_correlate_negation_is_minus_one_and_affine_invariance():
    a = _rand((3, 50), 2)
    am, asd = O.stat(a, 0, 50, 0, 3)
    b = (np.float32(1.0) - a).astype(np.float32)
    bm, bsd = O.stat(b, 0, 50, 0, 3)
    assert O.correlate(a, am, asd, b, bm, bsd, 0, 0) == pytest.approx(-1.0, abs=1e-6)
    c = (a * np.float32(0.25) + np.float32(0.5)).astype(np.float32)
    cm, csd = O.stat(c, 0, 50, 0, 3)
    assert O.correlate(a, am, asd, c, cm, csd, 0, 0) == pytest.approx(1.0, abs=1e-6)


def test_correlate_ring_rotation_is_alignment():
    """b holds the window rotated by r slots, bFrameOff = r undoes it (impl/MathUtil.scala:190)."""
    a = _rand((2, 31), 3)
    am, asd = O.stat(a, 0, 31, 0, 2)
    for r in (0, 1, 17, 30):
        b = np.roll(a, r, axis=1)
        bm, bsd = O.stat(b, 0, 31, 0, 2)
        assert O.correlate(a, am, asd, b, bm, bsd, r, 0) == pytest.approx(1.0, abs=1e-6)


def test_correlate_half_identical_halves_is_one():
    h = _rand((4, 20), 4)
    ring = np.concatenate([h, h], axis=1)
    for off in (0, 7, 39):
        assert O.correlate_half(4, 20, np.roll(ring, off, axis=1), off, 0) == pytest.approx(1.0, abs=1e-6)


def test_constant_window_is_not_finite():
    ring = np.full((1, 8), 0.5, dtype=np.float32)
    assert not np.isfinite(O.correlate_half(1, 4, ring, 0, 0))


def test_avg_returns_float_of_fp64_sum():
    b = _rand(1000, 5)
    assert O.avg(b, 10, 900) == np.float32(b[10:910].astype(np.float64).sum() / 900)


def test_normalize_is_fp32_and_unclipped():
    norm = np.array([[0.25, 0.75], [0.0, 2.0]], dtype=np.float32)
    b = np.array([[0.0, 0.5, 1.0], [1.0, 3.0, -1.0]], dtype=np.float32)
    out = O.normalize(norm, b, 0, 3)
    want = np.stack([(b[0] - norm[0, 0]) / (norm[0, 1] - norm[0, 0]), (b[1] - norm[1, 0]) / (norm[1, 1] - norm[1, 0])])
    assert np.array_equal(out, want.astype(np.float32))
    assert out[0, 0] < 0 and out[0, 2] > 1  # no clipping
    assert np.array_equal(O.normalize(None, b, 0, 3), b)


@pytest.mark.parametrize("n,want", [(0, 0), (255, 0), (256, 1), (767, 1), (768, 2), (44100, 86), (22050, 43)])
def test_full_to_feat_rounds_half_up(n, want):
    assert O.full_to_feat(n, 512) == want


def test_synth_generators_agree_bitwise():
    for f in (-1, 0, 7):
        assert np.array_equal(O.synth_file(SEED, f, 300, NUM_CH), O.synth_file_numpy(SEED, f, 300, NUM_CH))
    x = O.synth_file(SEED, 0, 5000, NUM_CH)
    assert x[:, 0].min() > 0  # loudness stays positive: its logarithm is taken


def test_boost_law():
    """DB copy with loudness scaled by g = d**0.6 must get boost 1/d."""
    L, ts, frames = 64, 9, 400
    for d in (0.5, 2.0):
        g = d ** 0.6
        plants = [dict(file=0, offset=50, src_file=-1, src_offset=ts, length=L, amp=0.0, gain=g)]
        db = synth_db(1, frames, plants)
        tgt = synth_target(ts, L)
        cfg = corr_config(ts, L, max_boost=100.0, num_matches=1)
        _, sims, boosts = O.correlation_run(cfg, tgt, norms_default(), db)
        assert boosts[0][50] == pytest.approx(1.0 / d, rel=1e-5)
        # MFCC rows are untouched and a scaled loudness row is perfectly correlated
        assert sims[0][50] == pytest.approx(1.0, abs=1e-6)


def test_planted_matches_are_found_in_noise_order():
    L, ts, nf, frames = 96, 17, 12, 900
    plants = make_plants(5, L, nf, frames, ts)
    db = synth_db(nf, frames, plants)
    tgt = synth_target(ts, L)
    got, _, _ = O.correlation_run(corr_config(ts, L, num_matches=5), tgt, norms_default(), db)
    assert [(m.file, m.start // STEP) for m in got] == [(p["file"], p["offset"]) for p in plants]
    sims = [m.sim for m in got]
    assert sims == sorted(sims, reverse=True)
    assert all(m.stop - m.start == L * STEP and m.boost_out == 1.0 for m in got)


def test_short_file_yields_single_stale_ring_window():
    """A file shorter than the target still yields one loop iteration (impl/FeatureCorrelationImpl.scala:190-195)."""
    assert O.num_windows(10, 64) == 1 and O.num_windows(0, 64) == 0 and O.num_windows(64, 64) == 1
    L, ts = 64, 3
    db = synth_db(2, [300, 10])
    tgt = synth_target(ts, L)
    _, sims, _ = O.correlation_run(corr_config(ts, L, num_matches=3), tgt, norms_default(), db)
    assert sims[1].shape == (1,)


def test_max_boost_gate_zeroes_sim():
    L, ts, frames = 64, 9, 400
    plants = [dict(file=0, offset=50, src_file=-1, src_offset=ts, length=L, amp=0.0, gain=0.05)]
    db = synth_db(1, frames, plants)
    tgt = synth_target(ts, L)
    _, sims, boosts = O.correlation_run(corr_config(ts, L, max_boost=2.0), tgt, norms_default(), db)
    assert boosts[0][50] > 2.0 and sims[0][50] == 0.0


def test_segmentation_breaks_ascending_and_spaced():
    frames = np.concatenate([O.synth_file(SEED, 0, 400, NUM_CH), O.synth_file(SEED, 1, 400, NUM_CH) * np.float32(0.5),
                             O.synth_file(SEED, 2, 400, NUM_CH)])
    cfg = O.SegmConfig(0, 0, 0, 0, 22050, 0.5, 1, 4, 40 * STEP, STEP, NUM_CH - 1)
    breaks, dense = O.segmentation_run(cfg, frames, norms_default())
    assert dense.shape[0] == frames.shape[0] - 2 * 43 + 1
    sims = [b.sim for b in breaks]
    assert sims == sorted(sims) and len(breaks) == 4
    pos = sorted(b.pos for b in breaks)
    assert all(q - p >= 40 * STEP for p, q in zip(pos, pos[1:]))


def test_selfsim_is_symmetric_with_unit_diagonal():
    frames = O.synth_file(SEED, 0, 200, NUM_CH)
    cfg = O.SelfSimConfig(0, 0, 0, 0, 4 * STEP, 3, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    ext, decim, af_start, num_corrs = O.selfsim_geometry(cfg, 200, 200)
    assert (decim, af_start, num_corrs, ext) == (3, 0, 200 - 8 + 1, (200 - 8 + 1) // 3)
    sim, img = O.selfsim_run(cfg, frames, None, norms_default())
    assert np.array_equal(sim, sim.T)
    assert np.allclose(np.diag(sim), 1.0, atol=1e-6)
    # y axis is flipped in the raster (impl/SelfSimilarityImpl.scala:152-155): the diagonal runs bottom-left to top-right
    assert all(img[ext - 1 - i, i] == 0xFFFFFF for i in range(ext))


def test_selfsim_auto_decimation_cap():
    cfg = O.SelfSimConfig(0, 0, 0, 0, 44100, 1, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    ext, decim, _, num_corrs = O.selfsim_geometry(cfg, 155039, 155039)
    assert num_corrs == 155039 - 172 + 1 and decim == 4 and ext == num_corrs // 4 == 38717


def test_feature_stats_percentiles_bracket_the_data():
    """FeatureStats (impl/FeatureStatsImpl.scala:56-140): the warped-histogram percentiles lie close to the true
    1st / 99th percentiles of each file, and the merge takes the outermost ones."""
    from common import synth_db
    db = synth_db(3, [4000, 2500, 1200])
    mm = O.feature_stats(db)
    per = [O.feature_stats([x]) for x in db]
    for ch in range(db[0].shape[1]):
        assert mm[ch, 0] == min(p[ch, 0] for p in per) and mm[ch, 1] == max(p[ch, 1] for p in per)
        for x, p in zip(db, per):
            lo, hi = np.percentile(x[:, ch], [1, 99])
            spread = x[:, ch].max() - x[:, ch].min()
            assert abs(p[ch, 0] - lo) < 0.02 * spread and abs(p[ch, 1] - hi) < 0.02 * spread


def test_crosssim_is_the_correlation_of_the_shorter_span():
    """CrossSimilarity (impl/CrossSimilarityImpl.scala:92-168, ring = target length): the curve equals the
    punch-in similarities of FeatureCorrelation with the shorter span as the target; argument order is free."""
    from common import STEP, corr_config, norms_default, synth_db, synth_target
    long_ = synth_db(1, 500)[0]
    short = long_[120:180].copy()
    norms = norms_default()
    a = O.crosssim_run(short, long_, norms, 0.5, 8.0)
    b = O.crosssim_run(long_, short, norms, 0.5, 8.0)
    assert len(a) == 500 - 60 + 1 and np.array_equal(a, b)
    assert abs(a[120] - 1.0) < 1e-6 and int(np.argmax(a)) == 120
    cfg = corr_config(0, 60, num_matches=1)
    _, sims, _ = O.correlation_run(cfg, short, norms, [long_])
    assert np.array_equal(a, sims[0])
    assert O.crosssim_span(None, None, 500, STEP) == (0, 500)
    assert O.crosssim_span(10 * STEP, 700 * STEP, 500, STEP) == (10, 490)
    assert O.crosssim_span(900 * STEP, None, 500, STEP) == (500, 0)
