"""End-to-end runs of the processors from files on disk (feature AIFFs, *_feat.xml, feat_norms.aif),
through the library's host side and the GPU, against the oracle fed with the same arrays."""
import os
import zlib

import numpy as np
import pytest

from common import NUM_CH, SEED, STEP, O, corr_config, make_plants, norms_default, synth_db, synth_target
from strugatzki_amd import processors as P
from strugatzki_amd.processors import FeatureCorrelation, FeatureExtraction, FeatureSegmentation, SelfSimilarity, Span

pytestmark = pytest.mark.gpu
RATE = 44100.0 / STEP


def _write_entry(folder, name, frames, audio_name=None, num_coeffs=NUM_CH - 1, fft=(1024, 2)):
    feat = os.path.join(folder, f"{name}_feat.aif")
    meta = os.path.join(folder, f"{name}_feat.xml")
    P.write_float_aiff(feat, frames, RATE)
    cfg = FeatureExtraction.Config(audio_name or os.path.join(folder, f"{name}.aif"), feat, meta, num_coeffs, fft[0],
                                   fft[1], 0)
    open(meta, "w").write(cfg.to_xml())
    return meta


def _write_norms(folder, norms):
    P.write_float_aiff(os.path.join(folder, "feat_norms.aif"), np.ascontiguousarray(norms.T), RATE)


def _make_db(tmp_path, nf=8, frames=700, L=64, ts=11, nplants=3):
    db_dir = str(tmp_path / "db")
    os.makedirs(db_dir)
    plants = make_plants(nplants, L, nf, frames, ts)
    db = synth_db(nf, frames, plants)
    tgt = synth_target(ts, L)
    for f, x in enumerate(db):
        _write_entry(db_dir, f"snd{f:03d}", x)
    # entries the folder scan must skip: other numCoeffs, other step size
    _write_entry(db_dir, "zz_other_coeffs", O.synth_file(SEED, 90, 300, 11), num_coeffs=10)
    _write_entry(db_dir, "zz_other_step", O.synth_file(SEED, 91, 300, NUM_CH), fft=(2048, 2))
    in_meta = _write_entry(db_dir, "input", tgt)  # the input lives in the same folder and must be excluded
    norms = norms_default()
    _write_norms(db_dir, norms)
    return db_dir, in_meta, db, tgt, norms, plants


@pytest.mark.parametrize("num_matches,num_per_file,min_spacing", [(5, 1, 0), (6, 2, 30 * STEP)])
def test_feature_correlation_from_files(ctx, tmp_path, num_matches, num_per_file, min_spacing, screen_kernel_env):
    L, ts = 64, 11
    db_dir, in_meta, db, tgt, norms, plants = _make_db(tmp_path, L=L, ts=ts)
    cfg = FeatureCorrelation.Config(db_dir, in_meta, FeatureCorrelation.Punch(Span(ts * STEP, (ts + L) * STEP), 0.5),
                                    None, 22050, 88200, True, 8.0, num_matches, num_per_file, min_spacing)
    seen = []
    got = FeatureCorrelation.run(ctx, cfg, lambda p: seen.append(p) or True)
    want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=num_matches, num_per_file=num_per_file,
                                               min_spacing=min_spacing), tgt, norms, db, dense=False)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g.file == os.path.join(db_dir, f"snd{w.file:03d}.aif")
        assert (g.punch.start, g.punch.stop) == (w.start, w.stop)
        assert abs(g.sim - w.sim) <= 1e-5 and g.boostOut == 1.0
    assert seen and seen[-1] == 1.0 and seen == sorted(seen)


def test_feature_correlation_keeps_the_database_resident_and_runs_on_several_contexts(tmp_path, screen_kernel_env):
    """A second query against the same folder finds the files on the GPU (no per-file progress of a load); touching
    a file loads again.  Two and three contexts (files dealt round robin) return the single-context list."""
    import strugatzki_amd as S
    L, ts = 64, 11
    db_dir, in_meta, db, tgt, norms, plants = _make_db(tmp_path, L=L, ts=ts)
    mk = lambda t0, nm, npf: FeatureCorrelation.Config(db_dir, in_meta, FeatureCorrelation.Punch(
        Span(t0 * STEP, (t0 + L) * STEP), 0.5), None, 22050, 88200, True, 8.0, nm, npf, 0)
    with S.Context(0) as c0:
        seen = []
        first = FeatureCorrelation.run(c0, mk(ts, 5, 1), lambda p: seen.append(p) or True)
        assert any(p < 0.5 for p in seen)  # the load reports one step per file
        tag = P.N.lib.sgz_db_tag(c0._h)
        assert tag and db_dir.encode() in tag
        seen = []
        again = FeatureCorrelation.run(c0, mk(ts, 5, 1), lambda p: seen.append(p) or True)
        assert seen == [0.6, 1.0] and again == first
        other = FeatureCorrelation.run(c0, mk(ts + 3, 5, 2))  # another query, same resident files
        want, _, _ = O.correlation_run(corr_config(ts + 3, L, num_matches=5, num_per_file=2), tgt, norms, db,
                                       dense=False)
        assert [(m.punch.start, m.punch.stop) for m in other] == [(w.start, w.stop) for w in want]
        os.utime(os.path.join(db_dir, "snd001_feat.aif"), ns=(1, 1))
        seen = []
        FeatureCorrelation.run(c0, mk(ts, 5, 1), lambda p: seen.append(p) or True)
        assert any(p < 0.5 for p in seen)  # a changed file: loaded anew
        for n in (2, 3):
            ctxs = [S.Context(0) for _ in range(n)]
            try:
                for nm, npf in ((5, 1), (6, 3)):
                    assert FeatureCorrelation.run(ctxs, mk(ts, nm, npf)) == FeatureCorrelation.run(c0, mk(ts, nm, npf))
                seen = []
                FeatureCorrelation.run(ctxs, mk(ts, 5, 1), lambda p: seen.append(p) or True)
                assert seen == [0.6, 1.0]
            finally:
                for c in ctxs:
                    c.close()


def test_feature_correlation_reads_a_changed_meta_file_again(tmp_path):
    """Parsed meta files are kept between calls (a query service scans the same folder for every query); one that
    changes on disk is parsed again: here the file with the best match drops out of the database because its meta
    file now names another number of coefficients (impl/FeatureCorrelationImpl.scala:50-55), then comes back."""
    import strugatzki_amd as S
    L, ts = 64, 11
    db_dir, in_meta, db, tgt, norms, plants = _make_db(tmp_path, L=L, ts=ts)
    cfg = FeatureCorrelation.Config(db_dir, in_meta, FeatureCorrelation.Punch(Span(ts * STEP, (ts + L) * STEP), 0.5),
                                    None, 22050, 88200, True, 8.0, 3, 1, 0)
    with S.Context(0) as c0:
        first = FeatureCorrelation.run(c0, cfg)
        assert FeatureCorrelation.run(c0, cfg) == first
        best = os.path.basename(first[0].file)[:-len(".aif")]
        meta = os.path.join(db_dir, best + "_feat.xml")
        text = open(meta).read()
        mc = FeatureExtraction.Config.from_xml(text)
        other = FeatureExtraction.Config(mc.audioInput, mc.featureOutput, mc.metaOutput, 10, mc.fftSize, mc.fftOverlap,
                                         mc.channelsBehavior)
        open(meta, "w").write(other.to_xml())
        without = FeatureCorrelation.run(c0, cfg)
        assert all(os.path.basename(m.file) != best + ".aif" for m in without) and without[0] == first[1]
        open(meta, "w").write(text)
        os.utime(meta, ns=(2_000_000_000, 2_000_000_000))  # (same size as at first: the time tells)
        assert FeatureCorrelation.run(c0, cfg) == first


@pytest.mark.parametrize("num_matches,num_per_file,min_punch,max_punch", [(4, 1, 20, 60), (6, 2, 5, 200)])
def test_feature_correlation_punch_out_from_files(ctx, tmp_path, num_matches, num_per_file, min_punch, max_punch):
    """punchIn + punchOut: spans run from the punch-in to the punch-out offset, sim = sqrt(inSim * outSim)."""
    from common import punch_config
    Lin, Lout, ts, gap = 48, 32, 11, 30
    db_dir, in_meta, db, _, norms, _ = _make_db(tmp_path, nf=6, frames=500, L=Lin + gap + Lout, ts=ts, nplants=3)
    tgt = synth_target(ts, Lin + gap + Lout)
    P.write_float_aiff(os.path.join(db_dir, "input_feat.aif"), tgt, RATE)
    o_start = ts + Lin + gap
    cfg = FeatureCorrelation.Config(db_dir, in_meta, FeatureCorrelation.Punch(Span(ts * STEP, (ts + Lin) * STEP), 0.5),
                                    FeatureCorrelation.Punch(Span(o_start * STEP, (o_start + Lout) * STEP), 0.25),
                                    min_punch * STEP, max_punch * STEP, True, 8.0, num_matches, num_per_file, 0)
    got = FeatureCorrelation.run(ctx, cfg)
    ocfg = punch_config(ts, Lin, o_start, Lout, min_punch=min_punch, max_punch=max_punch, out_weight=0.25,
                        num_matches=num_matches, num_per_file=num_per_file)
    want, _, _ = O.correlation_run_punch(ocfg, tgt, norms, db)
    assert len(got) == len(want) > 0
    for g, w in zip(got, want):
        assert g.file == os.path.join(db_dir, f"snd{w.file:03d}.aif")
        assert (g.punch.start, g.punch.stop) == (w.start, w.stop)
        assert abs(g.sim - w.sim) <= 1e-5 or (np.isnan(g.sim) and np.isnan(w.sim))
        assert abs(g.boostIn - w.boost_in) <= 2e-6 * w.boost_in and abs(g.boostOut - w.boost_out) <= 2e-6 * w.boost_out


def test_feature_correlation_errors(ctx, tmp_path):
    L, ts = 64, 11
    db_dir, in_meta, *_ = _make_db(tmp_path, nf=2, L=L, ts=ts)
    base = FeatureCorrelation.Config(db_dir, in_meta, FeatureCorrelation.Punch(Span(ts * STEP, (ts + L) * STEP), 0.5))
    os.remove(os.path.join(db_dir, "feat_norms.aif"))
    with pytest.raises(P.N.NativeError) as e:
        FeatureCorrelation.run(ctx, base)
    assert e.value.code == P.N.SGZ_E_IO
    base.normalize = False
    assert len(FeatureCorrelation.run(ctx, base)) == 1
    with pytest.raises(P.N.NativeError):  # span beyond the input feature file
        FeatureCorrelation.run(ctx, FeatureCorrelation.Config(db_dir, in_meta, FeatureCorrelation.Punch(
            Span(0, 10 ** 9), 0.5), normalize=False))
    with pytest.raises(P.N.NativeError):  # punch-out span beyond the input feature file
        FeatureCorrelation.run(ctx, FeatureCorrelation.Config(db_dir, in_meta, base.punchIn,
                                                              FeatureCorrelation.Punch(Span(0, 10 ** 9), 0.5),
                                                              normalize=False))
    aborted = FeatureCorrelation.Config(db_dir, in_meta, base.punchIn, normalize=False)
    with pytest.raises(P.N.NativeError) as e:
        FeatureCorrelation.run(ctx, aborted, lambda p: False)
    assert "aborted" in str(e.value)


@pytest.mark.parametrize("swap", [False, True])
def test_cross_similarity_from_files(ctx, tmp_path, swap):
    """CrossSimilarity: the similarity curve file against the oracle; either input may be the shorter one."""
    d = str(tmp_path)
    long_ = O.synth_file(SEED, 3, 900, NUM_CH)
    short = O.synth_file(SEED, 4, 400, NUM_CH)
    short[100:200] = long_[333:433]
    norms = norms_default()
    _write_norms(d, norms)
    m_long, m_short = _write_entry(d, "long", long_), _write_entry(d, "short", short)
    span_short, span_long = P.Span(95 * STEP, 210 * STEP), P.Span(20 * STEP, None)
    out = os.path.join(d, "xsim.aif")
    cfg = P.CrossSimilarity.Config(d, m_long if swap else m_short, m_short if swap else m_long, out, "aiff",
                                   span_long if swap else span_short, span_short if swap else span_long, 0.5, True,
                                   8.0)
    seen = []
    P.CrossSimilarity.run(ctx, cfg, lambda p: seen.append(p) or True)
    a0, an = O.crosssim_span(95 * STEP, 210 * STEP, 400, STEP)
    b0, bn = O.crosssim_span(20 * STEP, None, 900, STEP)
    want = O.crosssim_run(short[a0:a0 + an], long_[b0:b0 + bn], norms, 0.5, 8.0)
    got, rate = P.read_audio_file(out)
    assert got.shape == (len(want), 1) and abs(rate - RATE) < 1e-9
    np.testing.assert_allclose(got[:, 0], want, rtol=0, atol=1e-5)
    assert int(np.argmax(got[:, 0])) == int(np.argmax(want))
    assert seen and seen[-1] == 1.0
    # the other output types the reference's <outputType> can name (CrossSimilarity.scala:55, 213)
    for typ, name in (("wav", "xsim.wav"), ("next", "xsim.au"), ("ircam", "xsim.sf")):
        out2 = os.path.join(d, name)
        P.CrossSimilarity.run(ctx, P.CrossSimilarity.Config(d, cfg.metaInput1, cfg.metaInput2, out2, typ, cfg.span1,
                                                            cfg.span2, 0.5, True, 8.0))
        got2, rate2 = P.read_audio_file(out2)
        # (wave and next hold an integer rate, ircam a single-precision one)
        assert np.array_equal(got2, got) and rate2 == (float(np.float32(RATE)) if typ == "ircam" else round(RATE))
    with pytest.raises(P.N.NativeError):
        P.CrossSimilarity.run(ctx, P.CrossSimilarity.Config(d, m_short, m_long, out, "raw"))


def test_feature_stats(ctx, tmp_path):
    """FeatureStats on the GPU against the oracle: array level (per file and merged) and from files, and the
    norms file it writes is the one the searches read."""
    db = synth_db(5, [3000, 1800, 2600, 900, 4100])
    db[3] = db[3] * np.float32(1.7) - np.float32(0.2)  # one file with another range
    want = O.feature_stats(db)
    ctx.db_reset(NUM_CH)
    for x in db:
        ctx.db_add_file(x)
    got, per = ctx.feature_stats(per_file=True)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
    for f, x in enumerate(db):
        np.testing.assert_allclose(per[f], O.feature_stats([x]), rtol=1e-12, atol=0)
    d = str(tmp_path)
    paths = []
    for f, x in enumerate(db):
        paths.append(os.path.join(d, f"s{f}_feat.aif"))
        P.write_float_aiff(paths[-1], x, RATE)
    seen = []
    spans = P.FeatureStats.run(ctx, paths, lambda p: seen.append(p) or True)
    np.testing.assert_allclose(np.array(spans), want, rtol=1e-12, atol=0)
    assert seen and seen[-1] == 1.0
    norms_path = os.path.join(d, P.FeatureStats.NORMALIZE_NAME)
    P.FeatureStats.write_norms(norms_path, spans)
    frames, rate = P.read_audio_file(norms_path)
    assert frames.shape == (2, NUM_CH) and rate == 44100.0
    assert np.array_equal(frames, np.array(spans, dtype=np.float64).T.astype(np.float32))
    with pytest.raises(P.N.NativeError):
        P.FeatureStats.run(ctx, [os.path.join(d, "missing_feat.aif")])


def test_feature_segmentation_from_files(ctx, tmp_path):
    d = str(tmp_path)
    frames = np.concatenate([O.synth_file(SEED, k, 300, NUM_CH) * np.float32(1.0 - 0.25 * (k % 2)) for k in range(4)])
    meta = _write_entry(d, "piece", frames)
    norms = norms_default()
    _write_norms(d, norms)
    cfg = FeatureSegmentation.Config(d, meta, Span(40 * STEP, 1100 * STEP), 22050, 0.5, True, 5, 22050)
    got = FeatureSegmentation.run(ctx, cfg)
    ocfg = O.SegmConfig(1, 1, 40 * STEP, 1100 * STEP, 22050, 0.5, 1, 5, 22050, STEP, NUM_CH - 1)
    want, _ = O.segmentation_run(ocfg, frames, norms)
    assert [(b.sim, b.pos) for b in got] == [(w.sim, w.pos) for w in want]


def _read_png_rgb(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(raw):
        n = int.from_bytes(raw[pos:pos + 4], "big")
        typ = raw[pos + 4:pos + 8]
        data = raw[pos + 8:pos + 8 + n]
        assert zlib.crc32(typ + data) == int.from_bytes(raw[pos + 8 + n:pos + 12 + n], "big")
        if typ == b"IHDR":
            w, h = int.from_bytes(data[:4], "big"), int.from_bytes(data[4:8], "big")
            assert data[8:10] == b"\x08\x02"
        elif typ == b"IDAT":
            idat += data
        pos += 12 + n
    px = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert np.all(px[:, 0] == 0)
    return px[:, 1:].reshape(h, w, 3)


def test_self_similarity_png(ctx, tmp_path):
    d = str(tmp_path)
    frames = O.synth_file(SEED, 5, 230, NUM_CH)
    frames[120:160] = frames[30:70]
    meta = _write_entry(d, "piece", frames)
    norms = norms_default()
    _write_norms(d, norms)
    out = os.path.join(d, "sim.png")
    cfg = SelfSimilarity.Config(d, meta, None, out, Span.all(), 8 * STEP, 2, 0.5, SelfSimilarity.GRAY, 1.0, 1.0, False,
                                True)
    SelfSimilarity.run(ctx, cfg)
    ocfg = O.SelfSimConfig(0, 0, 0, 0, 8 * STEP, 2, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    _, want = O.selfsim_run(ocfg, frames, None, norms)
    rgb = _read_png_rgb(out)
    assert rgb.shape == (want.shape[0], want.shape[1], 3)
    dpx = np.abs(rgb[:, :, 2].astype(int) - (want & 0xFF))
    assert dpx.max() <= 1 and (dpx > 0).mean() < 0.01
    with pytest.raises(P.N.NativeError):
        SelfSimilarity.run(ctx, SelfSimilarity.Config(d, meta, None, out, colors="psycho"))
    # colors = psycho (the reference's default): the palette is the caller's; the library hands out the matrix in
    # bands of rows, inputs and geometry (impl/SelfSimilarityImpl.scala:64-91) worked out from the same document
    pcfg = SelfSimilarity.Config(d, meta, None, out, Span.all(), 8 * STEP, 2, 0.5, "psycho", 1.0, 1.0, False, True)
    want_sim, _ = O.selfsim_run(ocfg, frames, None, norms, want_img=False)
    ext = want_sim.shape[0]
    bands = [SelfSimilarity.rows(ctx, pcfg, a, b) for a, b in ((0, 40), (40, 41), (41, ext))]
    got = np.concatenate(bands, axis=0)
    assert got.shape == want_sim.shape
    np.testing.assert_allclose(got, want_sim, rtol=0, atol=1e-5)
    assert np.array_equal(SelfSimilarity.rows(ctx, pcfg), got)
    with pytest.raises(P.N.NativeError):
        SelfSimilarity.rows(ctx, pcfg, 0, ext + 1)
