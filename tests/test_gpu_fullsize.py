"""The GPU path at the full sizes of the benchmark configurations, checked through properties that do not
need the CPU oracle to run at that size (it would take hours): known answers, agreement of independent GPU
paths, locality, symmetry, and oracle checks on small samples of the big problem."""
import os
import sys

import numpy as np
import pytest

from common import NUM_CH, SEED, STEP, O, corr_config, norms_default

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


def _bounds_enclose_exact_everywhere(ctx, nf, windows_per_file):
    """The screening kernel's upper bound of every window of every file against the exact kernel's similarity of
    that window (fp64, every window): never below it, and never far above."""
    ctx.corr_run_screen_f16(0, nf)
    up = np.concatenate([s for s, _ in ctx.corr_read_dense_range(0, nf)])
    ctx.corr_run(0, nf)
    ex = np.concatenate([s for s, _ in ctx.corr_read_dense_range(0, nf)])
    assert up.shape == ex.shape == (nf * windows_per_file,)
    d = up.astype(np.float64) - ex
    assert np.isfinite(d).all() and d.min() >= 0.0 and d.max() < 5e-3


def test_config3_search_30s_target_vs_10h_database():
    """120 files x 300 s, 30 s target, top-10 (the benchmark workload): the planted copies come back in noise
    order at their offsets, through the search over bounds and through the general path; the bounds of the
    matrix-core kernel bracket the exact kernel's similarities; a window picked by the search, re-scored, equals
    the CPU oracle's value for that window."""
    import bench as B
    import strugatzki_amd as S
    from strugatzki_amd import _native as N
    L, frames, nf, nm = 2584, 25839, 120, 10
    plant_files = [int(round(j * (nf - 1) / (nm - 1))) for j in range(nm)]
    plants = B.plant_list(nm, L, plant_files, frames)
    npl = [N.Plant(p["file"], p["offset"], p["src_file"], p["src_offset"], p["length"], p["amp"], p["gain"])
           for p in plants]
    with S.Context(0) as ctx:
        ctx.db_reset(NUM_CH)
        for g in range(nf):
            ctx.db_add_synthetic(B.SEED, g, frames, npl)
        ctx.set_norms(B.norms_default())
        with S.Context(0) as c2:
            c2.db_reset(NUM_CH)
            c2.db_add_synthetic(B.SEED, B.TARGET_FILE, B.TARGET_START + L + 16, [])
            tfile = c2.db_read_file(0)
        target = tfile[B.TARGET_START:B.TARGET_START + L].copy()
        ctx.corr_set_target(target, 0.5, 8.0)
        assert ctx.screen_f16_available
        want = [(p["file"], p["offset"]) for p in plants]
        fast = S.search(ctx, S.SearchParams(nm, 1, 0, STEP))
        assert [(m.file, m.start // STEP) for m in fast] == want
        assert all(m.stop - m.start == L * STEP for m in fast)
        assert [m.sim for m in fast] == sorted((m.sim for m in fast), reverse=True)
        general = S.search(ctx, S.SearchParams(nm, 1, 0, STEP, screen_head=False))
        assert [(m.file, m.start, m.stop, m.sim, m.boost_in) for m in general] == \
               [(m.file, m.start, m.stop, m.sim, m.boost_in) for m in fast]
        # bounds vs the exact kernel on every window of every file
        _bounds_enclose_exact_everywhere(ctx, nf, frames - L + 1)
        # one matched window against the oracle (a single window: file cut to L frames)
        m = fast[3]
        raw = ctx.db_read_file(m.file)[m.start // STEP:m.start // STEP + L]
        cfg = corr_config(B.TARGET_START, L, num_matches=1)
        one, _, _ = O.correlation_run(cfg, tfile, B.norms_default(), [raw], dense=False)
        assert abs(one[0].sim - m.sim) <= 1.2e-7 and abs(one[0].boost_in - m.boost_in) <= 2e-6 * m.boost_in


def test_config5_one_rank_shard_60s_target_1500_files():
    """One rank's shard of the 8-GPU configuration at its real size: 60 s target (5168 frames: the staged channel
    of the matrix-core kernel is twice as large, one workgroup per CU) against 1500 files x 300 s (125 h, 2.17 GB
    of frames, 31 008 000 windows).  Known answer (planted copies in noise order at their offsets); the search
    over bounds, the general path and the exact fp64 path return the same list; the bounds bracket the exact
    similarities on whole files; a returned window equals the CPU oracle's value for that window."""
    import bench as B
    import strugatzki_amd as S
    from strugatzki_amd import _native as N
    L, frames, nf, nm = 5168, 25839, 1500, 10
    plant_files = [int(round(j * (nf - 1) / (nm - 1))) for j in range(nm)]
    plants = B.plant_list(nm, L, plant_files, frames)
    npl = [N.Plant(p["file"], p["offset"], p["src_file"], p["src_offset"], p["length"], p["amp"], p["gain"])
           for p in plants]
    with S.Context(0) as ctx:
        ctx.db_reset(NUM_CH)
        for g in range(nf):
            ctx.db_add_synthetic(B.SEED, g, frames, npl)
        ctx.set_norms(B.norms_default())
        with S.Context(0) as c2:
            c2.db_reset(NUM_CH)
            c2.db_add_synthetic(B.SEED, B.TARGET_FILE, B.TARGET_START + L + 16, [])
            tfile = c2.db_read_file(0)
        target = tfile[B.TARGET_START:B.TARGET_START + L].copy()
        ctx.corr_set_target(target, 0.5, 8.0)
        assert ctx.screen_f16_available
        assert sum(ctx.corr_num_windows(f) for f in range(nf)) == nf * (frames - L + 1) == 31_008_000
        fast = S.search(ctx, S.SearchParams(nm, 1, 0, STEP))
        assert [(m.file, m.start // STEP) for m in fast] == [(p["file"], p["offset"]) for p in plants]
        assert all(m.stop - m.start == L * STEP for m in fast)
        general = S.search(ctx, S.SearchParams(nm, 1, 0, STEP, screen_head=False))
        exact = S.search(ctx, S.SearchParams(nm, 1, 0, STEP, screen=False))
        key = lambda ms: [(m.file, m.start, m.stop) for m in ms]
        assert key(general) == key(fast) == key(exact)
        assert [m.sim for m in general] == [m.sim for m in fast]
        assert all(abs(a.sim - b.sim) <= 1e-5 for a, b in zip(exact, fast))
        # bounds vs the exact kernel on every window of every file of the shard
        _bounds_enclose_exact_everywhere(ctx, nf, frames - L + 1)
        # one matched window against the oracle (a single window: file cut to L frames)
        m = fast[6]
        raw = ctx.db_read_file(m.file)[m.start // STEP:m.start // STEP + L]
        one, _, _ = O.correlation_run(corr_config(B.TARGET_START, L, num_matches=1), tfile, B.norms_default(), [raw],
                                      dense=False)
        assert abs(one[0].sim - m.sim) <= 1.2e-7 and abs(one[0].boost_in - m.boost_in) <= 2e-6 * m.boost_in


def test_config2_extraction_of_one_hour_mono():
    """1 h of 44.1 kHz mono, fftSize 1024, hop 512, 13 MFCC + loudness: frame count of the reference's framing,
    finite output, and locality -- the frames of a 20 s excerpt starting on a hop boundary equal the oracle's
    frames for that excerpt (away from the excerpt's first frames, which see its start)."""
    import strugatzki_amd as S
    sr, n = 44100.0, 3600 * 44100
    rng = np.random.default_rng(7)
    t = np.arange(n, dtype=np.float32)
    pcm = (0.25 * np.sin(t * np.float32(2 * np.pi * 220.0 / sr)) * (1.0 + 0.5 * np.sin(t * np.float32(2 * np.pi / sr)))
           ).astype(np.float32)
    pcm += (0.05 * rng.standard_normal(n)).astype(np.float32)
    with S.Context(0) as ctx:
        feat = ctx.extract(pcm, sr, 1024, 2, 13, 0)
    assert feat.shape == (-(-n // 512) - 1, 14) and np.isfinite(feat).all()
    h0, nh = 100_000, 20 * 44100 // 512
    seg = pcm[h0 * 512:(h0 + nh + 2) * 512]
    want = O.extract(seg, sr, 1024, 2, 13, 0)
    k = 4  # frames of the excerpt that depend on samples before its start
    np.testing.assert_allclose(feat[h0 + k:h0 + nh], want[k:nh], rtol=0, atol=2e-5)


def test_config4_self_similarity_of_a_30_minute_file():
    """30 min of features, full matrix (auto-decimated to at most 0xB504 columns as the reference does): the
    raster is symmetric about the anti-diagonal layout the reference draws, its diagonal is white (a window
    equals itself), and a 24 x 24 corner equals the oracle's pixels."""
    import strugatzki_amd as S
    frames = int(1800 * 44100 / 512)
    x = O.synth_file(SEED, 21, frames, NUM_CH)
    norms = norms_default()
    cfg = O.SelfSimConfig(0, 0, 0, 0, 44100, 1, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    ext, decim, af_start, num_corrs = O.selfsim_geometry(cfg, frames, frames)
    assert ext > 30000 and decim >= 2
    half = O.full_to_feat(44100, STEP)  # halfWinLen, impl/SelfSimilarityImpl.scala:61
    with S.Context(0) as ctx:
        _, img = ctx.selfsim(x, None, norms, af_start, half, decim, ext, 0.5, want_sim=False)
    assert img.shape == (ext, ext)
    diag = img[ext - 1 - np.arange(ext), np.arange(ext)]
    assert np.all(diag == 0xFFFFFF)
    sample = np.random.default_rng(3).integers(0, ext, size=(20000, 2))
    px, py = sample[:, 0], sample[:, 1]
    assert np.array_equal(img[ext - 1 - py, px], img[ext - 1 - px, py])
    # a corner against the oracle: the same geometry cut to the first 24 columns (windows on the diagonal overlap)
    ncut = 24
    cut = x[:af_start + ncut * decim + 2 * half - 1]
    ccfg = O.SelfSimConfig(0, 0, 0, 0, 44100, decim, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    assert O.selfsim_geometry(ccfg, cut.shape[0], cut.shape[0])[:3] == (ncut, decim, af_start)
    _, want = O.selfsim_run(ccfg, cut, None, norms)
    assert np.array_equal(img[ext - ncut:, :ncut], want)
    # and a block far from the diagonal: the oracle's two-file form compares windows of its first file (x) with
    # windows of its second (y) for x <= y
    X0, Y0, nb = 5000, 31000, 40
    f1 = x[af_start + X0 * decim: af_start + (X0 + nb) * decim + 2 * half - 1]
    f2 = x[af_start + Y0 * decim: af_start + (Y0 + nb) * decim + 2 * half - 1]
    bcfg = O.SelfSimConfig(0, 0, 0, 0, 44100, decim, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    assert O.selfsim_geometry(bcfg, f1.shape[0], f2.shape[0])[:3] == (nb, decim, 0)
    _, blk = O.selfsim_run(bcfg, f1, f2, norms)
    for bx in range(nb):
        for by in range(bx, nb):
            assert img[ext - 1 - (Y0 + by), X0 + bx] == blk[nb - 1 - by, bx]
            assert img[ext - 1 - (X0 + bx), Y0 + by] == blk[nb - 1 - by, bx]  # the mirrored pixel
