"""GPU parity of the FeatureCorrelation path against the CPU oracle.

Bar: similarities and boosts within 1e-5 absolute of the oracle, match offsets
(punch spans, file indices) bit-exact.  The oracle itself is parity-unpinned
against the Scala reference (see oracle/strugatzki_oracle.c).
"""
import numpy as np
import pytest

from common import (NUM_CH, SEED, STEP, HipMem, O, corr_config, make_plants, native_plants, norms_default, synth_db,
                    synth_target)

pytestmark = pytest.mark.gpu

TOL = 1e-5  # absolute, on sim; stated by the north star


def _load(ctx, db, norms):
    ctx.db_reset(db[0].shape[1] if db else NUM_CH)
    for f in db:
        ctx.db_add_file(f)
    ctx.set_norms(norms)


def _gpu_dense(ctx, nfiles):
    ctx.corr_run(0, nfiles)
    return [ctx.corr_read_dense(f) for f in range(nfiles)]


def _check_dense(ctx, db, tgt, ts, L, norms, weight=0.5, max_boost=8.0):
    import strugatzki_amd as S
    cfg = corr_config(ts, L, weight=weight, normalize=norms is not None, max_boost=max_boost, num_matches=5)
    _, sims, boosts = O.correlation_run(cfg, tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], weight, max_boost)
    got = _gpu_dense(ctx, len(db))
    worst = 0.0
    for f, (gs, gb) in enumerate(got):
        n = ctx.corr_num_windows(f)
        # (a non-empty file shorter than the target has the reference's one window over stale ring contents)
        assert n == sims[f].shape[0] == gs.shape[0]
        np.testing.assert_allclose(gs, sims[f], rtol=0, atol=TOL)
        np.testing.assert_allclose(gb, boosts[f], rtol=2e-6, atol=0)
        if n:
            worst = max(worst, float(np.max(np.abs(gs - sims[f]))))
    return worst


@pytest.mark.parametrize("L,frames", [(64, 400), (431, 2583), (100, 1500), (16, 100), (33, 33)])
def test_dense_parity(ctx, L, frames):
    ts = 20
    nf = 4
    plants = make_plants(2, L, nf, frames, ts)
    db = synth_db(nf, frames, plants)
    tgt = synth_target(ts, L)
    worst = _check_dense(ctx, db, tgt, ts, L, norms_default())
    assert worst <= TOL


def test_dense_parity_ragged_and_short_files(ctx):
    L, ts = 120, 5
    # 119, 7 and 60 frames: shorter than the target -> one window over the ring's stale contents, which come from
    # the file before (1000 frames), from two files before, and from the still untouched ring (first file)
    frames = [60, 1000, 119, 7, 120, 121, 0, 2047, 1024 + 119, 1024 + 120, 1]
    db = synth_db(len(frames), frames)
    tgt = synth_target(ts, L)
    _check_dense(ctx, db, tgt, ts, L, norms_default())


@pytest.mark.parametrize("L,num_ch,normalize", [(2584, 14, True), (200, 14, True), (431, 14, False), (1, 14, True),
                                                 (1025, 5, True), (300, 64, True), (5168, 14, True), (77, 2, False)])
def test_target_statistics_from_the_device(ctx, L, num_ch, normalize):
    """readInBuffer (impl/FeatureCorrelationImpl.scala:83-98) runs on the device: mean / std-dev of the loudness row
    and of the other rows and ln(avg loudness) against MathUtil.stat / avg of the oracle.  The device adds in a fixed
    tree order, the reference sequentially: equal to a few units of the last place of a double."""
    ts = 3
    tgt = synth_target(ts, L, num_ch=num_ch)[ts:ts + L] * np.float32(1.7) + np.float32(0.125)
    norms = norms_default(num_ch) if normalize else None
    ctx.db_reset(num_ch)
    ctx.set_norms(norms)
    ctx.corr_set_target(tgt, 0.5, 8.0)
    got = ctx.corr_target_stats()
    mat = O.normalize(norms, np.ascontiguousarray(tgt.T), 0, L)  # [ch][frame]
    t_mean, t_std = O.stat(mat, 0, L, 0, 1)
    s_mean, s_std = O.stat(mat, 0, L, 1, num_ch - 1)
    ln_avg = np.log(np.float64(np.float32(O.avg(mat[0], 0, L))))
    want = np.array([t_mean, t_std, s_mean, s_std, ln_avg])
    np.testing.assert_allclose(got[[0, 2, 4]], want[[0, 2, 4]], rtol=1e-14, atol=0)
    # (a one-frame loudness row has no spread: the deviations are what rounding leaves)
    np.testing.assert_allclose(got[[1, 3]], want[[1, 3]], rtol=1e-13, atol=1e-300 if L > 1 else 1e-15)
    # the same target again gives the same bits (fixed summation order)
    ctx.corr_set_target(tgt, 0.5, 8.0)
    assert np.array_equal(ctx.corr_target_stats(), got)


def test_target_from_device_memory_is_the_target_from_host_memory(ctx):
    """sgz_corr_set_target_device: raw frames that are on the GPU already (a query's extraction leaves them there) give
    the statistics and the search of the same frames handed over in host memory, bit for bit."""
    import strugatzki_amd as S
    L, ts, nf, frames = 300, 9, 4, 2000
    db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
    tgt = np.ascontiguousarray(synth_target(ts, L)[ts:ts + L])
    ctx.db_reset(NUM_CH)
    for f in db:
        ctx.db_add_file(f)
    ctx.set_norms(norms_default())
    ctx.corr_set_target(tgt, 0.5, 8.0)
    want_stats = ctx.corr_target_stats()
    want = S.search(ctx, S.SearchParams(3, 1, 0, STEP))
    other = np.ascontiguousarray(synth_target(ts + 40, L)[ts + 40:ts + 40 + L])
    ctx.corr_set_target(other, 0.5, 8.0)  # (something else in between)
    hip = HipMem()
    try:
        dev = hip.alloc(tgt.nbytes)
        hip.upload(dev, tgt)
        ctx.corr_set_target_device(dev, L, 0.5, 8.0)
        assert np.array_equal(ctx.corr_target_stats(), want_stats)
        got = S.search(ctx, S.SearchParams(3, 1, 0, STEP))
    finally:
        ctx.synchronize()
        hip.free_all()
    assert [(g.sim, g.file, g.start, g.stop) for g in got] == [(w.sim, w.file, w.start, w.stop) for w in want]


@pytest.mark.parametrize("weight", [0.0, 1.0, 0.25])
def test_dense_parity_weights(ctx, weight):
    L, ts = 80, 11
    db = synth_db(3, 700, make_plants(1, L, 3, 700, ts))
    tgt = synth_target(ts, L)
    _check_dense(ctx, db, tgt, ts, L, norms_default(), weight=weight)


def test_dense_parity_without_normalisation(ctx):
    L, ts = 80, 11
    db = synth_db(3, 700, make_plants(1, L, 3, 700, ts))
    tgt = synth_target(ts, L)
    _check_dense(ctx, db, tgt, ts, L, None)


@pytest.mark.parametrize("weight", [0.5, 0.0, 1.0])
@pytest.mark.parametrize("normalize", [True, False])
def test_dense_parity_with_values_that_are_not_numbers(ctx, weight, normalize):
    """MathUtil.stat / correlate / avg run over one window's frames (impl/MathUtil.scala:29-62, 109-118, 177-196): a
    NaN or an infinity spoils exactly the windows that contain it, and what comes of it -- a NaN similarity, or 0
    where the boost gate sees a NaN -- depends on the channel group, the weight and the sign.  Every window against
    the oracle, NaN for NaN."""
    L, ts, frames = 90, 7, 1300
    db = synth_db(6, frames, make_plants(2, L, 6, frames, ts))
    db[0][500, 3] = np.nan           # an MFCC channel
    db[1][0, 0] = np.nan             # the loudness channel, first frame of the file
    db[1][frames - 1, 5] = np.inf    # last frame
    db[2][400, 0] = np.inf           # loudness +inf: the boost is 0, the gate open
    db[2][700, 0] = -np.inf          # loudness -inf: the boost is NaN, the gate shut
    db[3][640, 9] = -np.inf
    db[3][650, 0] = np.nan
    db[4][100:130, 1] = np.nan       # a run of them (file 5 holds none)
    tgt = synth_target(ts, L)
    norms = norms_default() if normalize else None
    cfg = corr_config(ts, L, weight=weight, normalize=normalize, num_matches=5)
    with np.errstate(all="ignore"):
        _, sims, boosts = O.correlation_run(cfg, tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], weight, 8.0)
    got = _gpu_dense(ctx, len(db))
    for f, (gs, gb) in enumerate(got):
        assert np.array_equal(np.isnan(gs), np.isnan(sims[f])), f"file {f}: NaN similarities in other windows"
        np.testing.assert_allclose(gs, sims[f], rtol=0, atol=TOL)
        np.testing.assert_allclose(gb, boosts[f], rtol=2e-6, atol=0)
    if weight == 0.5:  # (the windows around a value are spoilt, the rest of the file is not)
        assert np.isnan(sims[0][500 - L + 1:501]).all() and not np.isnan(sims[0][:500 - L + 1]).any() \
            and not np.isnan(sims[0][501:]).any()


def test_max_boost_gate_gives_zero_sim(ctx):
    """A quiet copy needs a boost above maxBoost: sim must be exactly 0f (impl/FeatureCorrelationImpl.scala:209)."""
    L, ts, frames = 64, 9, 600
    plants = [dict(file=0, offset=100, src_file=-1, src_offset=ts, length=L, amp=0.0, gain=0.05)]
    db = synth_db(2, frames, plants)
    tgt = synth_target(ts, L)
    _check_dense(ctx, db, tgt, ts, L, norms_default(), max_boost=2.0)
    sim, boost = ctx.corr_read_dense(0)
    assert boost[100] > 2.0 and sim[100] == 0.0
    # boost law of notes/BoostFactor.txt: loudness scaled by d**0.6 <-> boost 1/d; here loudness*0.05
    assert abs(boost[100] - 0.05 ** (-1 / 0.6)) / boost[100] < 1e-3


def test_long_target_two_lds_blocks(ctx):
    """L > 192*16 frames makes the kernel stage each channel in two blocks."""
    L, ts = 3300, 3
    db = synth_db(2, 5000, make_plants(1, L, 2, 5000, ts))
    tgt = synth_target(ts, L)
    _check_dense(ctx, db, tgt, ts, L, norms_default())


@pytest.mark.parametrize("num_matches,num_per_file,min_spacing", [(10, 1, 0), (5, 3, 0), (6, 2, 40 * STEP),
                                                                  (1, 1, 0), (12, 4, 3 * STEP)])
def test_search_matches_bit_exact(ctx, num_matches, num_per_file, min_spacing):
    import strugatzki_amd as S
    L, ts, nf, frames = 96, 17, 16, 900
    plants = make_plants(6, L, nf, frames, ts)
    db = synth_db(nf, frames, plants)
    tgt = synth_target(ts, L)
    norms = norms_default()
    cfg = corr_config(ts, L, num_matches=num_matches, num_per_file=num_per_file, min_spacing=min_spacing)
    want, _, _ = O.correlation_run(cfg, tgt, norms, db, dense=False)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    got = S.search(ctx, S.SearchParams(num_matches, num_per_file, min_spacing, STEP, capacity=1 << 16))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert (g.file, g.start, g.stop) == (w.file, w.start, w.stop)
        assert abs(g.sim - w.sim) <= TOL
        assert abs(g.boost_in - w.boost_in) <= 2e-6 * abs(w.boost_in)
        assert g.boost_out == 1.0


@pytest.mark.parametrize("L,frames", [(64, 400), (431, 2583), (3300, 5000)])
def test_screening_kernel_and_exact_rescoring(ctx, L, frames):
    """fp32-product screening stays within 1e-5 (in fact ~1e-7) of the oracle; re-scored candidates are exact."""
    ts, nf = 20, 3
    db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    norms = norms_default()
    _, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=3), tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    ctx.corr_run_screen(0, nf)
    for f in range(nf):
        gs, gb = ctx.corr_read_dense(f)
        np.testing.assert_allclose(gs, sims[f], rtol=0, atol=TOL)
        assert float(np.max(np.abs(gs - sims[f]))) < 2e-6
        np.testing.assert_allclose(gb, boosts[f], rtol=2e-6, atol=0)
    thr = 0.2
    cands = ctx.corr_compact(0, nf, thr, 1 << 16, rescore=True)
    assert len(cands) > 0
    for c in cands:
        want = sims[c["file"]][c["offset"]]
        assert abs(float(c["sim"]) - float(want)) <= 1.2e-7  # one fp32 ulp at most: same fp64 arithmetic as the oracle
        assert c["boost"] == pytest.approx(boosts[c["file"]][c["offset"]], rel=2e-6)


@pytest.mark.parametrize("L,frames", [(64, 400), (431, 2583), (3300, 5000), (16, 100), (33, 33), (100, 4500)])
def test_screening_f16_is_an_upper_bound(ctx, L, frames):
    """The matrix-core screening kernel writes sim + its error bound: never below the oracle, and close to it."""
    ts, nf = 20, 3
    db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    norms = norms_default()
    _, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=3), tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    assert ctx.screen_f16_available
    ctx.corr_run_screen_f16(0, nf)
    for f in range(nf):
        gs, gb = ctx.corr_read_dense(f)
        open_ = boosts[f] <= 8.0 * (1 - 1e-4)  # away from the boost gate
        d = gs.astype(np.float64) - sims[f]
        assert np.all(d[open_] >= 0.0)
        assert np.all(d[open_] <= 2e-2)
        np.testing.assert_allclose(gb, boosts[f], rtol=1e-5, atol=0)  # fp32 log / exp in the screening epilogue
    cands = ctx.corr_compact(0, nf, 0.2, 1 << 16, rescore=True)
    assert len(cands) > 0
    for c in cands:
        assert abs(float(c["sim"]) - float(sims[c["file"]][c["offset"]])) <= 1.2e-7


@pytest.mark.parametrize("nb", [4, 5, 6, 7, 9, 10])
def test_screening_f16_tile_widths(ctx, monkeypatch, nb):
    """The matrix-core kernel exists in several tile widths (256 * nb windows per workgroup; the library picks one
    per database and target, SGZ_HALF_NB forces one): same bounds, same search result for each.  Files of one
    window, of exactly one tile of the default width, one window more, and of several tiles."""
    L, ts = 431, 20
    frames = [L + 2 * 256 * nb + 77, L, L + 2047, L + 2048, 9000]
    nf = len(frames)
    db = [O.synth_file(SEED, g, frames[g], NUM_CH) for g in range(nf)]
    tgt = synth_target(ts, L)
    db[0][300:300 + L] = tgt[ts:ts + L] * np.float32(0.999) + db[0][300:300 + L] * np.float32(0.001)
    db[4][8000:8000 + L] = tgt[ts:ts + L]
    norms = norms_default()
    want, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=4), tgt, norms, db)
    monkeypatch.setenv("SGZ_HALF_NB", str(nb))
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    ctx.corr_run_screen_f16(0, nf)
    for f in range(nf):
        gs, gb = ctx.corr_read_dense(f)
        open_ = boosts[f] <= 8.0 * (1 - 1e-4)
        d = gs.astype(np.float64) - sims[f]
        assert gs.shape == sims[f].shape and np.all(d[open_] >= 0.0) and np.all(d[open_] <= 2e-2)
    import strugatzki_amd as S
    got = S.search(ctx, S.SearchParams(4, 1, 0, STEP))
    assert [(g.sim, g.file, g.start, g.stop) for g in got] == [(w.sim, w.file, w.start, w.stop) for w in want]
    monkeypatch.delenv("SGZ_HALF_NB")
    _load(ctx, db[:1], norms)  # (the next test's layout picks its own width again)


@pytest.mark.parametrize("L,frames", [(8000, 9300), (12498, 13400)])
def test_screening_f16_bound_at_long_targets(ctx, L, frames):
    """The fp32 summation part of the bound grows with the target length (an output element passes through
    numCh * ceil(S2 / 8) matrix instructions): the longest target the matrix-core kernel's LDS takes (12 498
    frames, 145 s) and one at 8 000 frames.  The data sit close to one level with a ripple of 1e-3, so that the
    cross terms are sums of many small products of both signs: the regime in which summation error is largest
    against the result."""
    ts, nf = 5, 2
    rng = np.random.default_rng(L)
    db = synth_db(nf, frames, make_plants(1, L, nf, frames, ts))
    norms = norms_default()
    ripple = (rng.standard_normal((frames, NUM_CH)) * 1e-3).astype(np.float32)
    db[1] = (np.float32(0.4) + ripple) * (norms[:, 1] - norms[:, 0]) + norms[:, 0]  # normalises to 0.4 + ripple
    tgt = synth_target(ts, L)
    _, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=2), tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    assert ctx.screen_f16_available
    ctx.corr_run_screen_f16(0, nf)
    for f in range(nf):
        gs, _ = ctx.corr_read_dense(f)
        open_ = np.isfinite(sims[f]) & (boosts[f] <= 8.0 * (1 - 1e-4))
        d = gs.astype(np.float64) - sims[f]
        assert open_.any() and np.all(d[open_] >= 0.0)
    import strugatzki_amd as S
    got = S.search(ctx, S.SearchParams(2, 1, 0, STEP))
    want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=2), tgt, norms, db, dense=False)
    assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]


def test_screening_f16_bound_with_values_in_the_subnormal_range(ctx):
    """Centred values below 6e-5 are subnormal in half precision and round with an absolute, not a relative
    error: files that sit within 1e-5 .. 1e-4 of their own mean level."""
    L, ts, nf, frames = 300, 11, 3, 2500
    rng = np.random.default_rng(5)
    norms = norms_default()
    db = synth_db(nf, frames, make_plants(1, L, nf, frames, ts))
    for f, amp in ((1, 3e-5), (2, 1e-4)):
        x = np.float32(0.5) + (rng.standard_normal((frames, NUM_CH)) * amp).astype(np.float32)
        db[f] = x * (norms[:, 1] - norms[:, 0]) + norms[:, 0]
    tgt = synth_target(ts, L)
    _, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=2), tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    ctx.corr_run_screen_f16(0, nf)
    for f in range(nf):
        gs, _ = ctx.corr_read_dense(f)
        open_ = np.isfinite(sims[f]) & (boosts[f] <= 8.0 * (1 - 1e-4))
        d = gs.astype(np.float64) - sims[f]
        assert open_.any() and np.all(d[open_] >= 0.0)
    import strugatzki_amd as S
    got = S.search(ctx, S.SearchParams(3, 1, 0, STEP))
    exact = S.search(ctx, S.SearchParams(3, 1, 0, STEP, screen=False))
    assert [(g.file, g.start, g.stop) for g in got] == [(e.file, e.start, e.stop) for e in exact]


def test_new_target_of_the_same_length_reuses_the_half_precision_preparation(ctx):
    """Image, prefix records and centring constants of the half-precision path depend on database, norms and
    target length only: a second target of the same length (searched without a new preparation) and a return to
    the first one give the oracle's lists; changing the norms or the database builds them anew."""
    import strugatzki_amd as S
    L, nf, frames = 120, 8, 1500
    norms = norms_default()
    db = synth_db(nf, frames, make_plants(3, L, nf, frames, 17))
    tgt = synth_target(17, 4 * L)
    _load(ctx, db, norms)
    for ts in (17, 200, 17):
        ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
        got = S.search(ctx, S.SearchParams(4, 1, 0, STEP))
        want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=4), tgt, norms, db, dense=False)
        assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
        assert all(abs(g.sim - w.sim) <= 1.2e-7 for g, w in zip(got, want))
    # general path: exact head run + half-precision tail run, twice (the tail's preparation is kept, the head's
    # exact run must not disturb it)
    for ts in (200, 17):
        ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
        got = S.search(ctx, S.SearchParams(4, 2, 0, STEP))
        want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=4, num_per_file=2), tgt, norms, db, dense=False)
        assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
    norms2 = norms.copy()
    norms2[:, 1] += np.float32(0.25)
    ctx.set_norms(norms2)
    ctx.corr_set_target(tgt[17:17 + L], 0.5, 8.0)
    got = S.search(ctx, S.SearchParams(4, 1, 0, STEP))
    want, _, _ = O.correlation_run(corr_config(17, L, num_matches=4), tgt, norms2, db, dense=False)
    assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]


def test_screening_f16_without_normalisation(ctx):
    """Features that are not normalised go through the matrix-core kernel as well (every file's image is centred
    on the file's own mean level); only values beyond the half-precision range do not: a target like that has no
    image (the search then screens with the fp32 kernel), a database file like that is marked and sent to the
    exact path.  Same matches as the oracle either way."""
    import strugatzki_amd as S
    from strugatzki_amd import _native as N
    L, ts, nf, frames = 431, 20, 6, 2583
    db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    cfg = corr_config(ts, L, num_matches=3, normalize=False)
    want, sims, boosts = O.correlation_run(cfg, tgt, None, db)
    _load(ctx, db, None)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    assert ctx.screen_f16_available
    ctx.corr_run_screen_f16(0, nf)
    for f in range(nf):
        gs, _ = ctx.corr_read_dense(f)
        open_ = boosts[f] <= 8.0 * (1 - 1e-4)
        d = gs.astype(np.float64) - sims[f]
        assert np.all(d[open_] >= 0.0) and np.all(d[open_] <= 2e-2)
    got = S.search(ctx, S.SearchParams(3, 1, 0, STEP))
    assert [(g.sim, g.file, g.start, g.stop) for g in got] == [(w.sim, w.file, w.start, w.stop) for w in want]
    # a database file beyond the half-precision range: marked, the search takes the exact path for the result
    big = [x.copy() for x in db]
    big[2] = big[2] * np.float32(3.0e5)
    want2, _, _ = O.correlation_run(cfg, tgt, None, big, dense=False)
    _load(ctx, big, None)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    got2 = S.search(ctx, S.SearchParams(3, 1, 0, STEP))
    assert [(g.sim, g.file, g.start, g.stop) for g in got2] == [(w.sim, w.file, w.start, w.stop) for w in want2]
    # a target beyond the range: no image, "auto" screens with the fp32 kernel
    tbig = tgt * np.float32(3.0e5)
    want3, _, _ = O.correlation_run(cfg, tbig, None, db, dense=False)
    _load(ctx, db, None)
    ctx.corr_set_target(tbig[ts:ts + L], 0.5, 8.0)
    assert not ctx.screen_f16_available
    with pytest.raises(N.NativeError):
        ctx.corr_run_screen_f16(0, nf)
    got3 = S.search(ctx, S.SearchParams(3, 1, 0, STEP))
    assert [(g.file, g.start, g.stop) for g in got3] == [(w.file, w.start, w.stop) for w in want3]


@pytest.mark.parametrize("screen", [False, "f32", "f16"])
def test_search_screened_equals_exact(ctx, screen):
    import strugatzki_amd as S
    L, ts, nf, frames = 96, 17, 24, 900
    plants = make_plants(8, L, nf, frames, ts)
    db = synth_db(nf, frames, plants)
    tgt = synth_target(ts, L)
    norms = norms_default()
    cfg = corr_config(ts, L, num_matches=6, num_per_file=2, min_spacing=10 * STEP)
    want, _, _ = O.correlation_run(cfg, tgt, norms, db, dense=False)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    got = S.search(ctx, S.SearchParams(6, 2, 10 * STEP, STEP, screen=bool(screen), screen_mode=screen or "auto"))
    assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
    for g, w in zip(got, want):
        assert abs(g.sim - w.sim) <= 1.2e-7


@pytest.mark.parametrize("num_ch", [2, 9, 21, 41])
def test_other_channel_counts(ctx, num_ch):
    """numCoeffs other than the default 13: the general preparation kernels (exact and half precision), dense
    parity, the upper bound, and the search over bounds against the oracle."""
    import strugatzki_amd as S
    L, ts, nf, frames = 120, 9, 5, 1500
    db = synth_db(nf, frames, make_plants(3, L, nf, frames, ts), num_ch=num_ch)
    tgt = synth_target(ts, L, num_ch=num_ch)
    norms = norms_default(num_ch)
    cfg = corr_config(ts, L, num_matches=3, num_per_file=1, num_ch=num_ch)
    want, sims, boosts = O.correlation_run(cfg, tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    for f, (gs, gb) in enumerate(_gpu_dense(ctx, nf)):
        assert float(np.max(np.abs(gs - sims[f]))) < 1e-5
    assert ctx.screen_f16_available
    ctx.corr_run_screen_f16(0, nf)
    for f in range(nf):
        gs, gb = ctx.corr_read_dense(f)
        open_ = boosts[f] <= 8.0 * (1 - 1e-4)
        assert np.all(gs[open_].astype(np.float64) >= sims[f][open_])
        np.testing.assert_allclose(gb, boosts[f], rtol=1e-5, atol=0)
    got = S.search(ctx, S.SearchParams(3, 1, 0, STEP))
    assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
    for g, w in zip(got, want):
        assert abs(g.sim - w.sim) <= 1.2e-7


@pytest.mark.parametrize("case", ["flat_stretches", "large_values", "offset_channels", "steps", "far_level"])
def test_bounds_hold_on_awkward_data(ctx, case):
    """The half-precision bounds on data built to stress them: stretches with almost no spread (the variance
    term of the bound), normalised values in the hundreds (half-precision range, large prefix sums in the fp32
    records), channel means far from the target's (the dm / sd term), level steps.  The upper bound stays
    above the oracle's similarity everywhere the boost gate is open, and the search over bounds returns the
    exact search's list."""
    import strugatzki_amd as S
    L, ts, nf, frames = 200, 11, 6, 3000
    rng = np.random.default_rng(99)
    db = synth_db(nf, frames, make_plants(3, L, nf, frames, ts))
    norms = norms_default()
    for f in range(nf):
        x = db[f]
        if case == "flat_stretches":
            for a in range(100, frames - 600, 700):
                x[a:a + 450] = x[a] + (rng.random((450, x.shape[1])) * 1e-4).astype(np.float32)
        elif case == "large_values":
            x *= np.float32(300.0)
            x[:, 0] = np.abs(x[:, 0]) + np.float32(1.0)
        elif case == "offset_channels":
            x[:, 1:] += np.linspace(0.0, 5.0, x.shape[1] - 1, dtype=np.float32)
        elif case == "steps":
            x[frames // 3:] += np.float32(2.0)
            x[2 * frames // 3:] -= np.float32(3.5)
            x[:, 0] = np.abs(x[:, 0]) + np.float32(0.05)
        elif case == "far_level" and f >= 2:
            # normalised values near 300 with a spread of 0.05: the fp32 normalisation itself rounds at a thousandth
            # of the spread
            r = (rng.standard_normal((frames, x.shape[1])) * 0.05).astype(np.float32)
            db[f] = (np.float32(300.0) + r) * (norms[:, 1] - norms[:, 0]) + norms[:, 0]
    tgt = synth_target(ts, L)
    cfg = corr_config(ts, L, num_matches=4, num_per_file=1)
    want, sims, boosts = O.correlation_run(cfg, tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    assert ctx.screen_f16_available
    ctx.corr_run_screen_f16(0, nf)
    for f in range(nf):
        gs, _ = ctx.corr_read_dense(f)
        ok = np.isfinite(sims[f]) & (boosts[f] <= 8.0 * (1 - 1e-4))
        assert np.all(gs[ok].astype(np.float64) >= sims[f][ok])
    got = S.search(ctx, S.SearchParams(4, 1, 0, STEP))
    exact = S.search(ctx, S.SearchParams(4, 1, 0, STEP, screen=False))
    from strugatzki_amd.engine import match_tuple
    assert [match_tuple(g) for g in got] == [match_tuple(e) for e in exact]
    assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]


def test_compact_top_leaves_out_files_below_the_kth_bound(ctx, screen_kernel_env):
    """sgz_corr_compact_top: T = the num_top-th largest per-file lower bound; only files that can reach it are
    collected (a subset of sgz_corr_compact_bounded's records, exactly re-scored), and the replay over them
    (sgz_sel_feed_top) gives the reference's matches.  More num_top than files: nothing is left out."""
    from strugatzki_amd.engine import Selector, match_tuple
    L, ts, nf, frames, k = 431, 20, 12, 2583, 3
    db = synth_db(nf, frames, make_plants(3, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    norms = norms_default()
    want, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=k, num_per_file=1), tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    ctx.corr_run_screen_f16(0, nf)
    every = ctx.corr_compact_bounded(0, nf, nf, 1 << 16)
    top, T = ctx.corr_compact_top(0, nf, k, 1 << 16)
    assert {(int(c["file"]), int(c["offset"])) for c in top} <= {(int(c["file"]), int(c["offset"])) for c in every}
    files = {int(f) for f in top["file"]}
    exact_max = sorted((float(sims[f].max()) for f in range(nf)), reverse=True)
    assert k <= len(files) < nf
    assert exact_max[k - 1] - 3e-3 <= T <= exact_max[k - 1]
    assert all(f in files for f in range(nf) if sims[f].max() >= T)
    for c in top:
        assert abs(float(c["sim"]) - float(sims[c["file"]][c["offset"]])) <= 1.2e-7
    sel = Selector(k, 1, 0, STEP, L)
    assert sel.feed_top(top, T)
    assert [match_tuple(m) for m in sel.result()] == [O.match_tuple(w) for w in want]
    sel.close()
    top2, T2 = ctx.corr_compact_top(0, nf, nf + 1, 1 << 16)
    assert T2 == float("-inf") and {int(f) for f in top2["file"]} == set(range(nf))


def test_compact_top_with_more_files_than_one_selection_round_holds(ctx):
    """1100 small files: the k-th largest bound is selected over more keys than the selection kernel keeps in
    registers (4 per thread of one workgroup); the threshold sits just below the k-th largest exact per-file
    maximum and the replay over the files kept equals the reference loop."""
    from strugatzki_amd.engine import Selector, match_tuple
    L, ts, nf, frames, k = 64, 5, 1100, 80, 7
    db = synth_db(nf, frames, make_plants(3, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    norms = norms_default()
    want, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=k, num_per_file=1), tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    ctx.corr_run_screen_f16(0, nf)
    top, T = ctx.corr_compact_top(0, nf, k, 1 << 16)
    exact_max = sorted((float(sims[f].max()) for f in range(nf)), reverse=True)
    assert exact_max[k - 1] - 3e-3 <= T <= exact_max[k - 1]
    files = {int(f) for f in top["file"]}
    assert all(f in files for f in range(nf) if sims[f].max() >= T) and len(files) < nf // 4
    sel = Selector(k, 1, 0, STEP, L)
    assert sel.feed_top(top, T)
    assert [match_tuple(m) for m in sel.result()] == [O.match_tuple(w) for w in want]
    sel.close()


@pytest.mark.parametrize("case", ["duplicate_head_files", "short_head_file", "single_file", "fewer_files_than_matches"])
def test_search_over_bounds_and_its_fallbacks(ctx, case, screen_kernel_env):
    """numPerFile = 1 with f16 screening searches over bounds; cases it cannot decide go to the exact path."""
    import strugatzki_amd as S
    L, ts, nf, frames = 96, 17, 12, 900
    fr = [frames] * nf
    num_matches = 5
    if case == "short_head_file":
        fr[2] = L - 1
    if case == "single_file":
        nf, fr = 1, [frames]
    if case == "fewer_files_than_matches":
        nf, fr = 3, [frames] * 3
    db = synth_db(nf, fr, make_plants(4, L, nf, frames, ts))
    if case == "duplicate_head_files":
        db[1] = db[0].copy()  # equal best sims collapse in the reference's sorted set
    tgt = synth_target(ts, L)
    norms = norms_default()
    cfg = corr_config(ts, L, num_matches=num_matches, num_per_file=1)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    want, _, _ = O.correlation_run(cfg, tgt, norms, db, dense=False)
    got = S.search(ctx, S.SearchParams(num_matches, 1, 0, STEP))
    assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
    for g, w in zip(got, want):
        assert abs(g.sim - w.sim) <= 1.2e-7


def _run_ranks(mode, world, backend, extra=()):
    import json
    import os
    import socket
    import subprocess
    import sys
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sharded_rank.py")
    procs = [subprocess.Popen([sys.executable, script, str(port), mode, str(r), str(world), backend, *map(str, extra)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    done = [pr.communicate(timeout=110) for pr in procs]
    # (a rank that dies takes the others down with "connection closed": show every rank's last words)
    assert all(pr.returncode == 0 for pr in procs), "\n".join(
        f"--- rank {r} (exit {pr.returncode}) ---\n{err[-1500:]}" for r, (pr, (_, err)) in enumerate(zip(procs, done)))
    for out, err in done:
        outs.append(json.loads(out.strip().splitlines()[-1]))
    return outs


def _sharded_want(mode=""):
    from common import SHARDED_CASE, sharded_case_db
    q = SHARDED_CASE
    db = sharded_case_db(mode)
    tgt = synth_target(q["ts"], q["L"])
    want, _, _ = O.correlation_run(corr_config(q["ts"], q["L"], num_matches=q["num_matches"], num_per_file=1), tgt,
                                   norms_default(), db, dense=False)
    return want


@pytest.mark.parametrize("mode", ["bounds", "general"])
def test_search_sharded_single_rank_rccl(mode):
    """The multi-GPU entry point with a one-rank RCCL group: candidate buffers on the GPU, one all-gather.
    Runs tests/sharded_rank.py as a child process."""
    want = _sharded_want()
    (got,) = _run_ranks(mode, 1, "nccl")
    assert [(g[0], g[1], g[2]) for g in got] == [(w.file, w.start, w.stop) for w in want]
    for g, w in zip(got, want):
        assert abs(g[3] - w.sim) <= 1.2e-7


@pytest.mark.parametrize("mode", ["bounds", "general", "exact", "headbounds", "nan"])
def test_search_sharded_two_ranks_on_one_gpu(mode, screen_kernel_env):
    """Two ranks (two processes sharing the one GPU of the box, gloo for the exchange): both end with the
    single-process result, with and without replicated head files ("headbounds": replicated head files in the
    search over bounds, whose windows every rank reports), and over a database with constant stretches
    ("nan": the bounds cannot decide, every rank switches to the exact maxima, the file that starts with a NaN
    window travels densely)."""
    want = _sharded_want(mode)
    outs = _run_ranks(mode, 2, "gloo")
    assert outs[0] == outs[1] or mode == "nan"  # (NaN != NaN in the printed lists)
    for out in outs:
        assert [(g[0], g[1], g[2]) for g in out] == [(w.file, w.start, w.stop) for w in want]
        for g, w in zip(out, want):
            assert abs(g[3] - w.sim) <= (1e-5 if mode in ("exact", "nan") else 1.2e-7) or (g[3] != g[3] and
                                                                                           w.sim != w.sim)
    if mode == "nan":
        assert any(w.sim != w.sim for w in want)  # the case does exercise a kept NaN match


def test_search_sharded_after_the_norms_change():
    """The ring contents the ranks hand their short files are derived under the norms in force: a second search with
    a target of the same length and other norms must not reuse them (impl/FeatureCorrelationImpl.scala:154,
    190-197: the untouched part of the ring holds zeros, i.e. raw values that normalise to 0)."""
    L, ts = 120, 5
    frames = [60, 30, 1000, 119, 700, 400, 260]
    db = synth_db(len(frames), frames)
    tgt = synth_target(ts, L)
    other = norms_default().copy()
    other[:, 0] -= np.float32(0.25)
    other[:, 1] += np.float32(0.5)
    outs = _run_ranks("norms", 2, "gloo")
    for k, norms in enumerate((norms_default(), other)):
        want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=5), tgt, norms, db, dense=False)
        for out in outs:
            got = out[k]
            assert [(g[0], g[1], g[2]) for g in got] == [(w.file, w.start, w.stop) for w in want]
            assert all(abs(g[3] - w.sim) <= 1.2e-7 for g, w in zip(got, want))


@pytest.mark.parametrize("world,seed", [(2, 5), (3, 6)])
def test_search_sharded_random_cases(world, seed, screen_kernel_env):
    """Random databases, parameters and ways of sharding (search over bounds, replicated head files, exact),
    ranks without any file among them: every rank's result equals the oracle's."""
    outs = _run_ranks("fuzz", world, "gloo", extra=(40, seed))
    assert all(o["bad"] == 0 and o["cases"] == 40 for o in outs), outs


@pytest.mark.parametrize("num_matches,num_per_file,min_spacing,min_punch,max_punch",
                         [(5, 1, 0, 20, 60), (6, 3, 25 * STEP, 0, 12), (3, 1, 0, 40, 40)])
def test_search_punch_out(ctx, num_matches, num_per_file, min_spacing, min_punch, max_punch):
    """Punch-in / punch-out pairing over the GPU's dense similarities of both targets, against the oracle."""
    import strugatzki_amd as S
    from common import punch_config
    Lin, Lout, ts, gap, nf, frames = 48, 32, 9, 20, 7, 420
    db = synth_db(nf, frames, make_plants(3, Lin + gap + Lout, nf, frames, ts))
    tgt = synth_target(ts, Lin + gap + Lout)
    norms = norms_default()
    o_start = ts + Lin + gap
    cfg = punch_config(ts, Lin, o_start, Lout, min_punch=min_punch, max_punch=max_punch, out_weight=0.75,
                       num_matches=num_matches, num_per_file=num_per_file, min_spacing=min_spacing)
    want, din, dout = O.correlation_run_punch(cfg, tgt, norms, db)
    _load(ctx, db, norms)
    got = S.search_punch(ctx, S.SearchParams(num_matches, num_per_file, min_spacing, STEP), tgt[ts:ts + Lin], 0.5,
                         tgt[o_start:o_start + Lout], 0.75, 8.0, min_punch, max_punch)
    assert len(want) > 0
    assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
    for g, w in zip(got, want):
        # (a negative inSim * outSim gives a NaN similarity, which the gate lets in while the set has room)
        assert abs(g.sim - w.sim) <= TOL or (np.isnan(g.sim) and np.isnan(w.sim))
        assert abs(g.boost_in - w.boost_in) <= 2e-6 * w.boost_in
        assert abs(g.boost_out - w.boost_out) <= 2e-6 * w.boost_out
    # the punch-out similarities the oracle computed are those of the dense run with the punch-out target
    sim_out, _ = ctx.corr_read_dense(2)
    have = ~np.isnan(dout[2][0][:len(sim_out)])
    assert have.any() and np.max(np.abs(sim_out[have] - dout[2][0][:len(sim_out)][have])) <= TOL


def test_search_falls_back_to_dense_on_candidate_overflow(ctx):
    import strugatzki_amd as S
    L, ts, nf, frames = 64, 4, 6, 500
    db = synth_db(nf, frames)
    tgt = synth_target(ts, L)
    norms = norms_default()
    cfg = corr_config(ts, L, num_matches=2, num_per_file=2)
    want, _, _ = O.correlation_run(cfg, tgt, norms, db, dense=False)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    got = S.search(ctx, S.SearchParams(2, 2, 0, STEP, capacity=1))
    assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]


def test_smaller_database_after_a_larger_one_with_unscreenable_file(ctx, screen_kernel_env):
    """A file the half-precision screening cannot handle (a constant stretch: NaN windows) is reported through the
    candidate count, without candidate records.  The re-scoring that follows must not read the slots of the
    candidate buffer, which still hold records of the earlier, larger database (file indices and frame pointers
    that are gone)."""
    import strugatzki_amd as S
    L, ts = 64, 4
    tgt = synth_target(ts, L)
    norms = norms_default()
    big = synth_db(40, 2500, make_plants(4, L, 40, 2500, ts))
    _load(ctx, big, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    for npf in (1, 3):  # both candidate paths leave records behind
        got = S.search(ctx, S.SearchParams(6, npf, 0, STEP, capacity=4096))
        want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=6, num_per_file=npf), tgt, norms, big, dense=False)
        assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
    small = synth_db(3, 700, make_plants(2, L, 3, 700, ts))
    small[1][100:400] = small[1][100]
    _load(ctx, small, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    for npf in (1, 3):
        want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=2, num_per_file=npf), tgt, norms, small,
                                       dense=False)
        got = S.search(ctx, S.SearchParams(2, npf, 0, STEP, capacity=4096, screen_mode="auto"))
        assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
        for g, w in zip(got, want):
            assert abs(g.sim - w.sim) <= TOL


def test_prefetch_of_dense_results(ctx):
    """sgz_corr_prefetch_dense: the read that follows returns what a plain read returns, also with more work queued
    behind the prefetch; a read of other files in between goes its own way; a new run over the files discards it."""
    L, ts, nf, frames = 100, 7, 6, 1500
    db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    _load(ctx, db, norms_default())
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    ctx.corr_run(0, nf)
    plain = ctx.corr_read_dense_range(0, nf)
    ctx.corr_run(0, 3)
    ctx.corr_prefetch_dense(0, 3)
    ctx.corr_run_screen_f16(3, nf)  # queued behind the prefetch
    other = ctx.corr_read_dense_range(1, 2)  # not the prefetched range
    got = ctx.corr_read_dense_range(0, 3)
    for f in range(3):
        assert np.array_equal(got[f][0], plain[f][0]) and np.array_equal(got[f][1], plain[f][1])
    assert np.array_equal(other[0][0], plain[1][0])
    # a second read is a plain one again; a run over the files in between makes a pending prefetch void
    again = ctx.corr_read_dense_range(0, 3)
    assert np.array_equal(again[2][0], plain[2][0])
    ctx.corr_prefetch_dense(0, 2)
    ctx.corr_set_target(tgt[ts + 1:ts + 1 + L], 0.5, 8.0)
    ctx.corr_run(0, 2)
    fresh = ctx.corr_read_dense_range(0, 2)
    ctx.corr_run(0, 2)
    assert np.array_equal(fresh[0][0], ctx.corr_read_dense_range(0, 2)[0][0])
    assert not np.array_equal(fresh[0][0], plain[0][0])


def test_synthetic_generator_matches_oracle_bits(ctx):
    L, ts, frames = 50, 8, 300
    plants = make_plants(2, L, 3, frames, ts)
    db = synth_db(3, frames, plants)
    ctx.db_reset(NUM_CH)
    for f in range(3):
        ctx.db_add_synthetic(SEED, f, frames, native_plants(plants))
    for f in range(3):
        assert np.array_equal(ctx.db_read_file(f), db[f])


def test_identity_window_scores_one(ctx):
    """correlate(a, a) == 1.0 (impl/MathUtil.scala:195): the target cut out of a database file."""
    L, frames = 200, 1200
    db = synth_db(1, frames)
    tgt = db[0][300:300 + L]
    _load(ctx, db, norms_default())
    ctx.corr_set_target(tgt, 0.5, 8.0)
    ctx.corr_run(0, 1)
    sim, boost = ctx.corr_read_dense(0)
    assert abs(sim[300] - 1.0) <= 1e-6 and abs(boost[300] - 1.0) <= 1e-6
    assert int(np.argmax(sim)) == 300


def test_kernel_time_history_and_sampling(ctx):
    """Every run carries timing events by default; sgz_set_kernel_timing thins them out or turns them off."""
    L, ts, nf, frames = 96, 17, 4, 3000
    db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
    _load(ctx, db, norms_default())
    ctx.corr_set_target(synth_target(ts, L)[ts:ts + L], 0.5, 8.0)
    before = len(ctx.kernel_ms_history(256))
    for _ in range(6):
        ctx.corr_run_screen_f16(0, nf)
    ctx.synchronize()
    h = ctx.kernel_ms_history(256)
    assert len(h) == min(before + 6, 256) and np.all(h[-6:] > 0) and np.all(h[-6:] < 50)
    assert abs(ctx.corr_last_kernel_ms() - h[-1]) < 1e-6
    ctx.set_kernel_timing(3)
    for _ in range(6):
        ctx.corr_run(0, nf)
    ctx.synchronize()
    assert len(ctx.kernel_ms_history(256)) == min(before + 8, 256)  # runs 0 and 3 of the six
    ctx.set_kernel_timing(0)
    ctx.corr_run(0, nf)
    ctx.synchronize()
    assert len(ctx.kernel_ms_history(256)) == min(before + 8, 256)
    ctx.set_kernel_timing(1)


@pytest.mark.parametrize("weight", [0.0, 0.5, 1.0])
def test_constant_stretches_in_the_exact_kernel(ctx, weight):
    """Windows over constant features (digital silence): the reference's two-pass statistics give a variance of
    exactly 0 and a NaN similarity, which is never offered.  The exact kernel's prefix-sum statistics cannot
    tell such a window from one with a tiny variance; it hands those windows to the two-pass kernels."""
    import strugatzki_amd as S
    L, ts, nf, frames = 64, 7, 3, 1200
    db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
    db[0][200:500] = db[0][200]             # everything constant
    db[1][100:400, 0] = np.float32(0.75)     # constant loudness only
    db[2][300:600, 1:] = np.float32(0.25)    # all MFCC channels at one value
    tgt = synth_target(ts, L)
    norms = norms_default()
    cfg = corr_config(ts, L, weight=weight, num_matches=5, num_per_file=2)
    want, sims, boosts = O.correlation_run(cfg, tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], weight, 8.0)
    nan_seen = 0
    for f, (gs, gb) in enumerate(_gpu_dense(ctx, nf)):
        assert np.array_equal(np.isnan(gs), np.isnan(sims[f]))
        assert np.array_equal(np.isinf(gs), np.isinf(sims[f]))
        ok = np.isfinite(sims[f])
        assert float(np.max(np.abs(gs[ok] - sims[f][ok]))) < 1e-5
        nan_seen += int(np.isnan(gs).sum())
    if weight > 0.0:
        assert nan_seen > 0
    for screen in (False, "f32", "auto"):
        got = S.search(ctx, S.SearchParams(5, 2, 0, STEP, screen=bool(screen), screen_mode=screen or "auto"))
        assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]


def test_tail_trim_shortens_every_file(ctx):
    """sgz_corr_set_tail_trim: window counts and values are those of the files without their last frames; a file
    that becomes shorter than the target keeps one window, an emptied one none; 0 restores whole files."""
    L, trim = 40, 25
    db = synth_db(4, [300, 60, 20, 140], [])
    tgt = synth_target(3, L)
    norms = norms_default()

    def dense(files, t):
        ctx.db_reset(NUM_CH)
        for f in files:
            ctx.db_add_file(f)
        ctx.set_norms(norms)
        ctx.corr_set_tail_trim(t)
        ctx.corr_set_target(tgt[3:3 + L], 0.5, 8.0)
        ctx.corr_run(0, len(files))
        counts = [ctx.corr_num_windows(f) for f in range(len(files))]
        return counts, ctx.corr_read_dense_range(0, len(files))

    counts, got = dense(db, trim)
    assert counts == [300 - trim - L + 1, 1, 0, 140 - trim - L + 1]
    cut_counts, want = dense([x[:max(x.shape[0] - trim, 0)] for x in db], 0)
    assert cut_counts == counts
    for (gs, gb), (ws, wb) in zip(got, want):
        assert np.array_equal(gs, ws, equal_nan=True) and np.array_equal(gb, wb, equal_nan=True)
    import strugatzki_amd as S
    with pytest.raises(S.NativeError):
        ctx.corr_set_tail_trim(-1)
    full_counts, _ = dense(db, 0)
    assert full_counts == [261, 21, 1, 101]


def test_database_files_as_audio_files_hold_them(ctx):
    """sgz_db_add_file_pcm: big-endian floats (what feature AIFFs hold), little-endian ones and integer layouts go up
    as bytes and are decoded on the device: the resident frames are bit for bit those of the decoded values."""
    num_ch = NUM_CH
    rng = np.random.default_rng(5)
    files = [O.synth_file(SEED, g, n, num_ch) for g, n in enumerate((1500, 1, 777))]
    ctx.db_reset(num_ch)
    for x in files:
        ctx.db_add_file_pcm(x.astype(">f4").tobytes(), 1, x.shape[0])
    ctx.db_add_file_pcm(files[0].astype("<f4").tobytes(), 0, files[0].shape[0])
    i16 = rng.integers(-32768, 32767, size=(333, num_ch)).astype(np.int16)
    ctx.db_add_file_pcm(i16.astype(">i2").tobytes(), 3, 333)
    ctx.db_add_file_pcm(b"", 1, 0)
    assert ctx.num_files == 6
    for f, want in enumerate(files + [files[0], i16.astype(np.float32) / np.float32(32768)]):
        got = ctx.db_read_file(f)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert ctx.file_frames(5) == 0
    import strugatzki_amd as S
    with pytest.raises(S.NativeError):
        ctx.db_add_file_pcm(b"\0" * 64, 9, 1)
