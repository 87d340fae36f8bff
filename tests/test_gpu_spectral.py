"""The spectral screening kernel (block transforms instead of direct sums) against the CPU oracle.

What it has to guarantee is what the matrix-core kernel guarantees: the value written for a window is never
below the oracle's similarity (impl/FeatureCorrelationImpl.scala:199-210), the per-file lower bound never above
the file's largest similarity, and every search over its bounds returns the oracle's matches with exactly
re-scored similarities.  The oracle itself is parity-unpinned against the Scala reference.
"""
import numpy as np
import pytest

from common import NUM_CH, SEED, STEP, O, corr_config, make_plants, norms_default, synth_db, synth_target

pytestmark = pytest.mark.gpu

# the spectral bounds are an order of magnitude tighter than the half-precision ones (2e-2 in their tests)
SLACK = 2e-3


@pytest.fixture(params=["q16", "f32"])
def spectral(ctx, monkeypatch, request):
    """The shared context with the spectral kernel forced, once with the database's and the target's spectra kept as
    16-bit integers and once as fp32; back to the library's own choices afterwards."""
    monkeypatch.delenv("SGZ_SCREEN_KERNEL", raising=False)
    monkeypatch.delenv("SGZ_FFT_N", raising=False)
    monkeypatch.delenv("SGZ_SPECTRA", raising=False)
    ctx.set_screen_kernel("spectral")
    ctx.set_spectra_format(request.param)
    ctx.spectra_format_wanted = request.param
    yield ctx
    ctx.set_screen_kernel("auto")
    ctx.set_spectra_format("auto")


def _load(ctx, db, norms):
    ctx.db_reset(db[0].shape[1] if db else NUM_CH)
    for f in db:
        ctx.db_add_file(f)
    ctx.set_norms(norms)


def _check_upper(ctx, sims, boosts, max_boost=8.0, slack=SLACK):
    worst = 0.0
    for f in range(len(sims)):
        gs, gb = ctx.corr_read_dense(f)
        assert gs.shape == sims[f].shape
        open_ = np.isfinite(sims[f]) & (boosts[f] <= max_boost * (1 - 1e-4))  # away from the boost gate
        d = gs.astype(np.float64) - sims[f]
        if open_.any():
            assert np.all(d[open_] >= 0.0), f"file {f}: upper bound below the oracle by {d[open_].min():.3e}"
            worst = max(worst, float(d[open_].max()))
        fin = np.isfinite(boosts[f])
        np.testing.assert_allclose(gb[fin], boosts[f][fin], rtol=1e-5, atol=0)
    assert worst <= slack
    return worst


@pytest.mark.parametrize("L,n,extra", [
    (64, 4096, [0, 1, 500]),          # a target far shorter than the transform
    (431, 4096, [0, 1, 2 * 3666 + 7]),
    (3072, 4096, [0, 1, 1500]),       # the longest target a 4096-point block takes (at most 1025 windows per block)
    (1200, 8192, [0, 1, 300]),
    (2584, 16384, [0, 1]),            # the headline target length
])
def test_bounds_enclose_the_oracle(spectral, monkeypatch, L, n, extra):
    """Every transform length: files of one window, of a partly filled block, of exactly one block of windows plus
    `extra` (0, 1: the file then needs a second block, and its windows are dealt evenly over the two), a file
    shorter than the target (one window over what earlier files left in the ring) -- upper bounds, boosts, the k-th
    per-file lower bound, the search."""
    import strugatzki_amd as S
    ctx = spectral
    monkeypatch.setenv("SGZ_FFT_N", str(n))
    ts = 20
    hop = n - L + 1
    frames = [L, L + 700, max(L // 2, 1)] + [L - 1 + hop + e for e in extra]
    nf = len(frames)
    db = [O.synth_file(SEED, g, frames[g], NUM_CH) for g in range(nf)]
    tgt = synth_target(ts, L)
    db[1][300:300 + L] = tgt[ts:ts + L] * np.float32(0.999) + db[1][300:300 + L] * np.float32(0.001)
    db[3][hop - 5:hop - 5 + L] = tgt[ts:ts + L] * np.float32(0.98) + db[3][hop - 5:hop - 5 + L] * np.float32(0.02)
    if len(extra) > 1:  # a copy whose window is the only one of the file's second block
        db[4][hop:hop + L] = tgt[ts:ts + L] * np.float32(0.99) + db[4][hop:hop + L] * np.float32(0.01)
    norms = norms_default()
    k = 3
    want, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=k), tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    ctx.corr_run_screen_f16(0, nf)
    assert ctx.screen_kernel_in_use == "spectral"
    assert ctx.spectra_format_in_use == ctx.spectra_format_wanted
    gn, ghop, gblocks = ctx.screen_geometry
    assert (gn, ghop) == (n, hop)
    assert gblocks == sum(-(-len(s) // hop) for s in sims)
    _check_upper(ctx, sims, boosts)
    top, T = ctx.corr_compact_top(0, nf, k, 1 << 16)
    exact_max = sorted((float(np.nanmax(s)) for s in sims if len(s)), reverse=True)
    assert exact_max[k - 1] - SLACK <= T <= exact_max[k - 1]
    for c in top:
        assert abs(float(c["sim"]) - float(sims[c["file"]][c["offset"]])) <= 1.2e-7
    got = S.search(ctx, S.SearchParams(k, 1, 0, STEP))
    assert [(g.sim, g.file, g.start, g.stop) for g in got] == [(w.sim, w.file, w.start, w.stop) for w in want]


@pytest.mark.parametrize("n", [4096, 8192, 16384])
def test_a_new_target_for_a_resident_database(spectral, monkeypatch, n):
    """A query service sets a target per search.  With the layout in place the launch that transforms the new target
    also carries the second pass of its preparation (centring, statistics): the statistics must be the bits the
    stand-alone preparation gives (first target, no layout yet), and every search must return the oracle's matches."""
    import strugatzki_amd as S
    ctx = spectral
    monkeypatch.setenv("SGZ_FFT_N", str(n))
    L, nf, frames = 700, 5, 6000
    starts = [20, 900, 1800, 20]
    tfile = O.synth_file(SEED, -1, 2600, NUM_CH)
    db = synth_db(nf, frames)
    for k, ts in enumerate(starts[:3]):  # a copy of every target somewhere
        db[k + 1][500 + 300 * k:500 + 300 * k + L] = tfile[ts:ts + L] * np.float32(0.97) + \
            db[k + 1][500 + 300 * k:500 + 300 * k + L] * np.float32(0.03)
    norms = norms_default()
    _load(ctx, db, norms)
    first_stats = None
    for ts in starts:
        ctx.corr_set_target(tfile[ts:ts + L], 0.5, 8.0)
        stats = ctx.corr_target_stats()
        if ts == starts[0]:
            if first_stats is None:
                first_stats = stats      # prepared before any layout existed
            else:
                assert np.array_equal(stats, first_stats)  # the same target again, now beside its transform
        want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=3), tfile, norms, db, dense=False)
        got = S.search(ctx, S.SearchParams(3, 1, 0, STEP))
        assert ctx.screen_kernel_in_use == "spectral" and ctx.screen_geometry[0] == n
        assert [(g.sim, g.file, g.start, g.stop) for g in got] == [(w.sim, w.file, w.start, w.stop) for w in want]


@pytest.mark.parametrize("L,n,num_ch", [(700, 4096, 14), (3000, 4096, 14), (1500, 8192, 5), (2584, 16384, 14),
                                         (5168, 16384, 14), (900, 16384, 3)])
def test_bounds_with_a_target_set_on_a_standing_layout(spectral, monkeypatch, L, n, num_ch):
    """The second target of a length goes through other kernels than the first (its transform is cut into sixteen
    sub-transforms per channel pair, with one to six terms in the first stage here, and an odd channel count leaves
    the last pair half empty): the bounds it gives must enclose the oracle's similarities like the first one's."""
    ctx = spectral
    monkeypatch.setenv("SGZ_FFT_N", str(n))
    nf, frames = 3, L + 2500
    tfile = O.synth_file(SEED, -1, 2 * L + 100, num_ch)
    db = [O.synth_file(SEED, g, frames, num_ch) for g in range(nf)]
    norms = norms_default(num_ch)
    _load(ctx, db, norms)
    for ts in (7, L + 50):  # first target: the layout is built behind it; second: beside a standing layout
        db_t = tfile[ts:ts + L]
        cfg = corr_config(ts, L, num_matches=2, num_ch=num_ch)
        _, sims, boosts = O.correlation_run(cfg, tfile, norms, db)
        ctx.corr_set_target(db_t, 0.5, 8.0)
        ctx.corr_run_screen_f16(0, nf)
        assert ctx.screen_kernel_in_use == "spectral" and ctx.screen_geometry[0] == n
        _check_upper(ctx, sims, boosts)


@pytest.mark.parametrize("weight", [0.0, 1.0, 0.25])
def test_weights(spectral, weight):
    """temporalWeight 0 and 1 leave one channel group out (impl/FeatureCorrelationImpl.scala:201-206)."""
    import strugatzki_amd as S
    ctx = spectral
    L, ts, nf, frames = 300, 11, 3, 5000
    db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    norms = norms_default()
    cfg = corr_config(ts, L, weight=weight, num_matches=2)
    want, sims, boosts = O.correlation_run(cfg, tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], weight, 8.0)
    ctx.corr_run_screen_f16(0, nf)
    assert ctx.screen_kernel_in_use == "spectral"
    _check_upper(ctx, sims, boosts)
    got = S.search(ctx, S.SearchParams(2, 1, 0, STEP))
    assert [(g.sim, g.file, g.start, g.stop) for g in got] == [(w.sim, w.file, w.start, w.stop) for w in want]


@pytest.mark.parametrize("num_ch", [2, 3, 9, 14, 22])
def test_other_channel_counts(spectral, num_ch):
    """Channels are transformed in pairs (two real sequences per complex transform): even and odd counts, and the
    smallest one (loudness + one coefficient)."""
    import strugatzki_amd as S
    ctx = spectral
    L, ts, nf, frames = 200, 9, 3, 4600
    db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts), num_ch=num_ch)
    tgt = synth_target(ts, L, num_ch=num_ch)
    norms = norms_default(num_ch)
    cfg = corr_config(ts, L, num_matches=2, num_ch=num_ch)
    want, sims, boosts = O.correlation_run(cfg, tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    ctx.corr_run_screen_f16(0, nf)
    assert ctx.screen_kernel_in_use == "spectral"
    _check_upper(ctx, sims, boosts)
    got = S.search(ctx, S.SearchParams(2, 1, 0, STEP))
    assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
    assert all(abs(g.sim - w.sim) <= 1.2e-7 for g, w in zip(got, want))


def test_without_normalisation(spectral):
    import strugatzki_amd as S
    ctx = spectral
    L, ts, nf, frames = 431, 20, 4, 5200
    db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    cfg = corr_config(ts, L, num_matches=3, normalize=False)
    want, sims, boosts = O.correlation_run(cfg, tgt, None, db)
    _load(ctx, db, None)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    ctx.corr_run_screen_f16(0, nf)
    assert ctx.screen_kernel_in_use == "spectral"
    _check_upper(ctx, sims, boosts)
    got = S.search(ctx, S.SearchParams(3, 1, 0, STEP))
    assert [(g.sim, g.file, g.start, g.stop) for g in got] == [(w.sim, w.file, w.start, w.stop) for w in want]


@pytest.mark.parametrize("case", ["flat_stretches", "large_values", "offset_channels", "steps", "ripple", "far_level"])
def test_bounds_hold_on_awkward_data(spectral, case):
    """Data built to stress the bound: stretches with almost no spread (a window's 1 / sd is large against the
    block's norm, which the error term is proportional to), values in the hundreds, channel levels far from the
    target's, level steps inside a block (the block norm is dominated by frames outside the window), and files
    that sit at one level with a ripple of 1e-3 (cross terms are sums of many small products of both signs), and
    files whose level is thousands of times their spread (the normalisation's own rounding shows)."""
    import strugatzki_amd as S
    from strugatzki_amd.engine import match_tuple
    ctx = spectral
    L, ts, nf, frames = 200, 11, 6, 5000
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
        elif case == "ripple" and f >= 3:
            r = (rng.standard_normal((frames, NUM_CH)) * 1e-3).astype(np.float32)
            db[f] = (np.float32(0.4) + r) * (norms[:, 1] - norms[:, 0]) + norms[:, 0]
        elif case == "far_level" and f >= 2:
            # normalised values near 300 with a spread of 0.05: the fp32 normalisation rounds at 3e-5, a thousandth
            # of the spread, before the file's level is taken off
            r = (rng.standard_normal((frames, NUM_CH)) * 0.05).astype(np.float32)
            db[f] = (np.float32(300.0) + r) * (norms[:, 1] - norms[:, 0]) + norms[:, 0]
    tgt = synth_target(ts, L)
    cfg = corr_config(ts, L, num_matches=4, num_per_file=1)
    want, sims, boosts = O.correlation_run(cfg, tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    ctx.corr_run_screen_f16(0, nf)
    assert ctx.screen_kernel_in_use == "spectral"
    # (how far above the oracle the bound sits depends on the data here: only its side is checked)
    _check_upper(ctx, sims, boosts, slack=np.inf)
    got = S.search(ctx, S.SearchParams(4, 1, 0, STEP))
    exact = S.search(ctx, S.SearchParams(4, 1, 0, STEP, screen=False))
    assert [match_tuple(g) for g in got] == [match_tuple(e) for e in exact]
    assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]


def test_flat_stretches_do_not_widen_their_neighbours_bounds(spectral):
    """A search's launch takes the error terms g and u of a window as their largest values over the block's windows;
    windows without spread (g huge or not finite) are kept out of those maxima and made certain candidates instead.
    So next to flat stretches the ordinary windows keep bounds of the same order as in a file without any: what the
    block's maxima can add is capped (2e-3 of a channel group's weight through g, 2.5e-4 through u; windows that
    overlap a stretch in part lie below the caps and do enter the maxima), where a flat window's own g is of order 10."""
    ctx = spectral
    L, ts, nf, frames = 200, 11, 2, 5000
    rng = np.random.default_rng(7)
    plain = synth_db(nf, frames, ())
    flat = [x.copy() for x in plain]
    starts = list(range(100, frames - 600, 700))
    for x in flat:
        for a in starts:
            x[a:a + 450] = x[a] + (rng.random((450, x.shape[1])) * 1e-4).astype(np.float32)
    # windows that touch no flat stretch: [a + 450, a + 500] behind every stretch
    ordinary = np.array([o for a in starts for o in range(a + 450, a + 501)])
    norms = norms_default()
    tgt = synth_target(ts, L)
    cfg = corr_config(ts, L, num_matches=4, num_per_file=1)
    widths = []
    for db in (plain, flat):
        _, sims, boosts = O.correlation_run(cfg, tgt, norms, db)
        _load(ctx, db, norms)
        ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
        ctx.corr_run_screen_f16(0, nf)
        assert ctx.screen_kernel_in_use == "spectral"
        _check_upper(ctx, sims, boosts, slack=np.inf)
        d = np.concatenate([ctx.corr_read_dense(f)[0].astype(np.float64)[ordinary] - sims[f][ordinary] for f in range(nf)])
        assert np.all(d >= 0.0)
        widths.append(float(np.max(d)))
    assert widths[0] < 2e-3
    assert widths[1] < widths[0] + 2.5e-3, widths


@pytest.mark.parametrize("num_matches,num_per_file,min_spacing", [(6, 1, 0), (6, 2, 10 * STEP), (5, 3, 0)])
def test_search_is_the_same_with_either_kernel(ctx, monkeypatch, num_matches, num_per_file, min_spacing):
    """numPerFile = 1 (search over bounds) and the general path (exact head, screened tail over a file range that
    does not start at file 0): spectral, matrix-core and exact searches return the oracle's list."""
    import strugatzki_amd as S
    from strugatzki_amd.engine import match_tuple
    monkeypatch.delenv("SGZ_SCREEN_KERNEL", raising=False)
    monkeypatch.delenv("SGZ_FFT_N", raising=False)
    L, ts, nf, frames = 150, 17, 20, 4200
    db = synth_db(nf, frames, make_plants(8, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    norms = norms_default()
    cfg = corr_config(ts, L, num_matches=num_matches, num_per_file=num_per_file, min_spacing=min_spacing)
    want, _, _ = O.correlation_run(cfg, tgt, norms, db, dense=False)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    lists = {}
    try:
        for kern in ("spectral", "matrix"):
            ctx.set_screen_kernel(kern)
            lists[kern] = S.search(ctx, S.SearchParams(num_matches, num_per_file, min_spacing, STEP))
            assert ctx.screen_kernel_in_use == kern
    finally:
        ctx.set_screen_kernel("auto")
    lists["exact"] = S.search(ctx, S.SearchParams(num_matches, num_per_file, min_spacing, STEP, screen=False))
    for name, got in lists.items():
        assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want], name
        assert all(abs(g.sim - w.sim) <= 1.2e-7 for g, w in zip(got, want)), name
    assert [match_tuple(g) for g in lists["spectral"]] == [match_tuple(g) for g in lists["matrix"]]


def test_choice_of_kernel_and_transform_length(ctx, monkeypatch):
    """Left to itself the library takes the matrix cores for targets below 1024 frames and the spectral kernel
    above, with the transform length that moves the fewest bytes among those whose blocks are at least a quarter
    windows; the setting and the environment variable override it."""
    monkeypatch.delenv("SGZ_SCREEN_KERNEL", raising=False)
    monkeypatch.delenv("SGZ_FFT_N", raising=False)
    ctx.set_screen_kernel("auto")
    norms = norms_default()
    db = synth_db(2, 9000)
    _load(ctx, db, norms)
    tgt = synth_target(0, 4000)

    def in_use(L):
        ctx.corr_set_target(tgt[:L], 0.5, 8.0)
        ctx.corr_run_screen_f16(0, 2)
        return ctx.screen_kernel_in_use, ctx.screen_geometry[0]

    assert in_use(500) == ("matrix", 0)
    # 8001 windows per file: one 16384-point block (16384 values) against two 8192-point blocks or five of 4096
    assert in_use(1000) == ("matrix", 0)
    assert in_use(1024)[0] == "spectral"
    kern, n = in_use(3500)
    assert kern == "spectral" and n in (8192, 16384)  # 4096-point blocks would hold 597 windows: too few
    monkeypatch.setenv("SGZ_SCREEN_KERNEL", "matrix")
    ctx.set_norms(norms)  # (same norms: nothing to rebuild)
    ctx.set_screen_kernel("spectral")
    assert in_use(3500) == ("matrix", 0)
    monkeypatch.delenv("SGZ_SCREEN_KERNEL")
    ctx.set_screen_kernel("matrix")
    assert in_use(3500) == ("matrix", 0)
    ctx.set_screen_kernel("spectral")
    assert in_use(500)[0] == "spectral"
    ctx.set_screen_kernel("auto")
    assert in_use(500) == ("matrix", 0)


def test_new_target_of_the_same_length_reuses_the_spectra(spectral):
    """The blocks' spectra depend on database, norms and target length only: a second target of the same length,
    a return to the first, then other norms (spectra rebuilt) -- each time the oracle's list."""
    import strugatzki_amd as S
    ctx = spectral
    L, nf, frames = 120, 6, 4500
    norms = norms_default()
    db = synth_db(nf, frames, make_plants(3, L, nf, frames, 17))
    tgt = synth_target(17, 4 * L)
    _load(ctx, db, norms)
    for ts in (17, 200, 17):
        ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
        got = S.search(ctx, S.SearchParams(4, 1, 0, STEP))
        assert ctx.screen_kernel_in_use == "spectral"
        want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=4), tgt, norms, db, dense=False)
        assert [(g.sim, g.file, g.start, g.stop) for g in got] == [(w.sim, w.file, w.start, w.stop) for w in want]
    norms2 = norms.copy()
    norms2[:, 1] += np.float32(0.25)
    ctx.set_norms(norms2)
    ctx.corr_set_target(tgt[17:17 + L], 0.5, 8.0)
    got = S.search(ctx, S.SearchParams(4, 1, 0, STEP))
    want, _, _ = O.correlation_run(corr_config(17, L, num_matches=4), tgt, norms2, db, dense=False)
    assert [(g.sim, g.file, g.start, g.stop) for g in got] == [(w.sim, w.file, w.start, w.stop) for w in want]


def test_device_transform_equals_the_host_statement_of_it(ctx):
    """The cross terms the kernel forms are circular correlations: for a database file that holds the target at a
    known offset and a target with temporalWeight 1, the similarity bound at that offset is within the bound's
    own width of 1, and bounds elsewhere within it of the oracle -- the transform itself is checked bit for bit
    against numpy on the host in tests/test_fft_host.py."""
    L, ts, frames = 1500, 3, 9000
    db = synth_db(1, frames)
    tgt = synth_target(ts, L)
    db[0][4000:4000 + L] = tgt[ts:ts + L]
    norms = norms_default()
    _, sims, boosts = O.correlation_run(corr_config(ts, L, weight=1.0, num_matches=1), tgt, norms, db)
    ctx.set_screen_kernel("spectral")
    try:
        _load(ctx, db, norms)
        ctx.corr_set_target(tgt[ts:ts + L], 1.0, 8.0)
        ctx.corr_run_screen_f16(0, 1)
        gs, _ = ctx.corr_read_dense(0)
    finally:
        ctx.set_screen_kernel("auto")
    assert sims[0][4000] == pytest.approx(1.0, abs=1e-6)
    assert 0.0 <= gs[4000] - sims[0][4000] <= SLACK
    open_ = boosts[0] <= 8.0 * (1 - 1e-4)
    d = gs.astype(np.float64) - sims[0]
    assert np.all(d[open_] >= 0) and d[open_].max() <= SLACK


def test_candidate_lists_instead_of_dense_bounds(spectral):
    """sgz_corr_set_screen_output(1): the spectral kernel lists the windows whose upper bound reaches their block's
    largest lower bound, and the search over bounds filters that list instead of scanning dense upper bounds.  Same
    records as from a dense run (for the k-th bound filter and for every file), dense readers refuse, a dense run
    afterwards serves them again, and two list runs in a row (the second list's counter was not cleared by a
    consumer) are both complete."""
    import strugatzki_amd as S
    from strugatzki_amd import _native as N
    ctx = spectral
    L, ts, nf, frames, k = 300, 20, 9, 5200, 3
    db = synth_db(nf, frames, make_plants(3, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    norms = norms_default()
    want, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=k), tgt, norms, db)
    _load(ctx, db, norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    ctx.corr_run_screen_f16(0, nf)
    dense_top, dense_T = ctx.corr_compact_top(0, nf, k, 1 << 16)
    dense_all = ctx.corr_compact_bounded(0, nf, nf, 1 << 16)
    key = lambda a: sorted((int(c["file"]), int(c["offset"]), float(c["sim"]), float(c["boost"])) for c in a)
    try:
        ctx.set_screen_output(True)
        for _ in range(2):
            ctx.corr_run_screen_f16(0, nf)
        top, T = ctx.corr_compact_top(0, nf, k, 1 << 16)
        every = ctx.corr_compact_bounded(0, nf, nf, 1 << 16)
        assert T == dense_T and key(top) == key(dense_top) and key(every) == key(dense_all)
        with pytest.raises(N.NativeError):
            ctx.corr_read_dense(0)
        with pytest.raises(N.NativeError):
            ctx.corr_compact(0, nf, 0.1, 1 << 16)
        ctx.corr_run_screen_f16(0, nf)
        top2, T2 = ctx.corr_compact_top(0, nf, k, 1 << 16)
        assert T2 == dense_T and key(top2) == key(dense_top)
    finally:
        ctx.set_screen_output(False)
    ctx.corr_run_screen_f16(0, nf)
    gs, gb = ctx.corr_read_dense(0)
    assert np.all(gs.astype(np.float64) >= sims[0])
    np.testing.assert_allclose(gb, boosts[0], rtol=1e-5)
    got = S.search(ctx, S.SearchParams(k, 1, 0, STEP))
    assert [(g.sim, g.file, g.start, g.stop) for g in got] == [(w.sim, w.file, w.start, w.stop) for w in want]


def test_candidate_list_overflow_falls_back(spectral):
    """Files whose frames repeat with period 2: every second window of a file has the file's largest similarity
    (to the last bit or within the bound), so the list of windows that reach their block's bound holds thousands
    of records per block and overflows.  The search then takes its other paths and still returns the oracle's
    matches; so does a database where only one file is like that."""
    import strugatzki_amd as S
    ctx = spectral
    L, ts, frames = 300, 7, 16000
    norms = norms_default()
    tgt = synth_target(ts, L)
    pair = O.synth_file(SEED, 50, 2, NUM_CH)
    periodic = np.tile(pair, (frames // 2, 1)).astype(np.float32)
    plain = synth_db(3, frames, make_plants(2, L, 3, frames, ts))
    for db in ([periodic, periodic.copy() * np.float32(0.9) + np.float32(0.01)], [plain[0], periodic, plain[1], plain[2]]):
        for k, npf in ((3, 1), (2, 1)):
            want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=k, num_per_file=npf), tgt, norms, db, dense=False)
            _load(ctx, db, norms)
            ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
            got = S.search(ctx, S.SearchParams(k, npf, 0, STEP))
            assert ctx.screen_kernel_in_use == "spectral"
            assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
            assert all(abs(g.sim - w.sim) <= 1.2e-7 for g, w in zip(got, want))
    # the list itself: more records than it holds -> the compaction reports an overflow, as for too small a buffer
    from strugatzki_amd import _native as N
    _load(ctx, [periodic, periodic], norms)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    try:
        ctx.set_screen_output(True)
        ctx.corr_run_screen_f16(0, 2)
        with pytest.raises(N.NativeError) as err:
            ctx.corr_compact_bounded(0, 2, 2, 1 << 16)
        assert err.value.code == N.SGZ_E_OVERFLOW
    finally:
        ctx.set_screen_output(False)


def test_sixteen_bit_spectra_widen_the_bounds_only_slightly(ctx, monkeypatch):
    """The same run with fp32 and with 16-bit spectra: both enclose the oracle (checked above), the 16-bit ones
    carry what the rounding to 16 bits was measured to be -- about half as much again; and the format can be
    queried and is rebuilt on change."""
    for v in ("SGZ_SCREEN_KERNEL", "SGZ_FFT_N", "SGZ_SPECTRA"):
        monkeypatch.delenv(v, raising=False)
    L, ts, nf, frames = 1500, 30, 3, 16000
    db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    norms = norms_default()
    _, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=2), tgt, norms, db)
    ctx.set_screen_kernel("spectral")
    try:
        _load(ctx, db, norms)
        ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
        width = {}
        for fmt in ("f32", "q16", "auto"):
            ctx.set_spectra_format(fmt)
            ctx.corr_run_screen_f16(0, nf)
            assert ctx.spectra_format_in_use == ("q16" if fmt == "auto" else fmt)
            width[fmt] = _check_upper(ctx, sims, boosts)
        assert width["f32"] <= width["q16"] <= 3.0 * width["f32"] + 1e-5
        assert width["auto"] == width["q16"]
        with pytest.raises(Exception):
            ctx.set_spectra_format("f16")
    finally:
        ctx.set_screen_kernel("auto")
        ctx.set_spectra_format("auto")


@pytest.mark.parametrize("value", [np.nan, np.inf, -np.inf])
@pytest.mark.parametrize("num_per_file", [1, 2])
def test_values_that_are_not_numbers(spectral, value, num_per_file):
    """A NaN or an infinity among a file's frames (feature files do not normally hold such values: the extraction
    replaces NaNs, impl/NonRealtimeProcessor.scala:178-190).  The reference's statistics run over one window's frames
    (impl/MathUtil.scala:29-62), so only the windows that contain the value are spoilt: the screened search, the
    exact search and the oracle's loop must return the same matches, whatever form the spectra are kept in -- a band
    of 16-bit spectra with such a component gets a NaN scale, the block's bounds come out as NaN and the file goes
    to the exact path, which evaluates every window of such a file from its frames."""
    import strugatzki_amd as S
    ctx = spectral
    L, ts, nf, frames = 300, 12, 4, 5000
    db = synth_db(nf, frames, make_plants(3, L, nf, frames, ts))
    db[1][2000, 3] = np.float32(value)
    db[2][10, 0] = np.float32(value)
    db[2][4990, 7] = np.float32(value)
    tgt = synth_target(ts, L)
    _load(ctx, db, norms_default())
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    got = S.search(ctx, S.SearchParams(4, num_per_file, 0, STEP))
    assert ctx.screen_kernel_in_use == "spectral"
    exact = S.search(ctx, S.SearchParams(4, num_per_file, 0, STEP, screen=False))
    with np.errstate(all="ignore"):
        want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=4, num_per_file=num_per_file), tgt,
                                       norms_default(), db, dense=False)
    assert len(want) == 4
    for res in (got, exact):
        assert [(g.file, g.start, g.stop) for g in res] == [(w.file, w.start, w.stop) for w in want]
        for g, w in zip(res, want):
            assert abs(g.sim - w.sim) <= 1.2e-7 or (np.isnan(g.sim) and np.isnan(w.sim))
