"""GPU parity of FeatureSegmentation / SelfSimilarity (MathUtil.correlateHalf) against the oracle."""
import numpy as np
import pytest

from common import NUM_CH, SEED, STEP, O, norms_default

pytestmark = pytest.mark.gpu


def _frames(n, seed_file=3):
    a = O.synth_file(SEED, seed_file, n, NUM_CH)
    b = O.synth_file(SEED, seed_file + 1, n, NUM_CH) * np.float32(0.6)
    k = n // 2
    return np.concatenate([a[:k], b[k:]])


@pytest.mark.parametrize("span,corr_len,weight,normalize", [
    ((None, None), 22050, 0.5, True),
    ((100 * STEP, 900 * STEP), 22050, 0.5, True),
    ((None, 700 * STEP), 11000, 0.0, True),
    ((33 * STEP, None), 5000, 1.0, False),
    ((None, None), 44100, 0.3, True),
])
def test_segmentation_curve_and_breaks(ctx, span, corr_len, weight, normalize):
    from strugatzki_amd.engine import segm_select
    frames = _frames(1200)
    norms = norms_default() if normalize else None
    cfg = O.SegmConfig(int(span[0] is not None), int(span[1] is not None), span[0] or 0, span[1] or 0, corr_len,
                       weight, int(normalize), 6, 22050, STEP, NUM_CH - 1)
    want, dense = O.segmentation_run(cfg, frames, norms)
    H = O.full_to_feat(corr_len, STEP)
    a0 = max(0, O.full_to_feat(span[0], STEP)) if span[0] is not None else 0
    a1 = min(frames.shape[0], O.full_to_feat(span[1], STEP)) if span[1] is not None else frames.shape[0]
    got = ctx.segm_curve(frames, norms, a0, a1, H, weight)
    assert got.shape == dense.shape
    np.testing.assert_allclose(got, dense, rtol=0, atol=1e-5)
    assert np.array_equal(got, dense)  # the kernel keeps the reference's summation order
    br = segm_select(got, a0, H, STEP, 6, 22050)
    assert [(b.sim, b.pos) for b in br] == [(w.sim, w.pos) for w in want]


@pytest.mark.parametrize("normalize", [True, False])
def test_segmentation_window_longer_than_span_gives_one_value(ctx, normalize):
    """The reference's loop still runs once: the span's frames followed by the zeros of its fresh buffer
    (impl/FeatureSegmentationImpl.scala:86, 107-113)."""
    frames = _frames(60)
    norms = norms_default() if normalize else None
    cfg = O.SegmConfig(1, 0, 4 * STEP, 0, 43 * STEP, 0.5, int(normalize), 3, 0, STEP, NUM_CH - 1)
    want, dense = O.segmentation_run(cfg, frames, norms)
    got = ctx.segm_curve(frames, norms, 4, 60, 43, 0.5)
    assert dense.shape == (1,) and np.array_equal(got, dense)
    assert ctx.segm_curve(frames, norms, 20, 20, 43, 0.5).shape == (0,)


@pytest.mark.parametrize("decim,corr_len,warp,ceil,inv", [(1, 4 * STEP, 1.0, 1.0, False), (3, 8 * STEP, 0.5, 0.8, True),
                                                         (2, 44100, 2.0, 1.0, False), (1, 6 * STEP, 1.0, 0.004, False),
                                                         (1, 2 * STEP, 0.05, 1.0, True)])
@pytest.mark.parametrize("kernel,tw", [("limbs", 0.5), ("f32", 0.5), ("limbs", 0.0), ("limbs", 1.0)])
def test_selfsim_matrix_and_image(ctx, monkeypatch, decim, corr_len, warp, ceil, inv, kernel, tw):
    """Both tile kernels (exact integer products of three 8-bit digits per value; plain fp32, SGZ_SELFSIM_KERNEL=f32)
    and the one-group cases (temporal weight 0 or 1: the other channel group is not read)."""
    frames = _frames(260)
    norms = norms_default()
    cfg = O.SelfSimConfig(0, 0, 0, 0, corr_len, decim, tw, 0, warp, ceil, int(inv), 1, STEP, NUM_CH - 1)
    ext, d, af_start, _ = O.selfsim_geometry(cfg, frames.shape[0], frames.shape[0])
    want_sim, want_img = O.selfsim_run(cfg, frames, None, norms)
    H = O.full_to_feat(corr_len, STEP)
    if kernel == "f32":
        monkeypatch.setenv("SGZ_SELFSIM_KERNEL", "f32")
    sim, img = ctx.selfsim(frames, None, norms, af_start, H, d, ext, tw, 0, warp, ceil, inv)
    np.testing.assert_allclose(sim, want_sim, rtol=0, atol=1e-5)
    assert np.array_equal(sim, sim.T)
    assert np.array_equal(img, want_img)  # pixels are integer work: bit-exact
    assert np.array_equal(img & 0xFF, (img >> 8) & 0xFF) and np.array_equal(img & 0xFF, (img >> 16) & 0xFF)


@pytest.mark.parametrize("kernel", ["limbs", "f32"])
def test_selfsim_image_when_the_fix_list_overflows(ctx, monkeypatch, kernel):
    """More uncertain pixels than their list holds (here: a list of 3 entries, SGZ_SELFSIM_FIX_CAP): the whole
    raster is evaluated in the reference's order instead.  (The integer kernel's own error bounds leave next to no
    pixel uncertain at this size: SGZ_SELFSIM_SLACK widens them.)"""
    frames = _frames(300)
    norms = norms_default()
    cfg = O.SelfSimConfig(0, 0, 0, 0, 2 * STEP, 1, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    ext, d, af_start, _ = O.selfsim_geometry(cfg, frames.shape[0], frames.shape[0])
    want_sim, want_img = O.selfsim_run(cfg, frames, None, norms)
    monkeypatch.setenv("SGZ_SELFSIM_FIX_CAP", "3")
    if kernel == "f32":
        monkeypatch.setenv("SGZ_SELFSIM_KERNEL", "f32")
    else:
        monkeypatch.setenv("SGZ_SELFSIM_SLACK", "1e-4")
    sim, img = ctx.selfsim(frames, None, norms, af_start, 2, d, ext, 0.5)
    assert np.array_equal(img, want_img)
    assert np.array_equal(sim, want_sim)  # (then the similarities are the reference-order ones as well)


@pytest.mark.parametrize("slack", ["1e-4", "3e-3"])
def test_selfsim_uncertain_pixels_listed_by_the_tile_kernel(ctx, monkeypatch, slack):
    """With widened error bounds (SGZ_SELFSIM_SLACK) a few per cent of the pixels -- with the larger value most of them,
    more than a tile collects before it appends them one by one -- go through the list and the exact evaluation:
    raster unchanged."""
    frames = _frames(700, 5)
    norms = norms_default()
    cfg = O.SelfSimConfig(0, 0, 0, 0, 8 * STEP, 1, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    ext, d, af_start, _ = O.selfsim_geometry(cfg, frames.shape[0], frames.shape[0])
    want_sim, want_img = O.selfsim_run(cfg, frames, None, norms)
    monkeypatch.setenv("SGZ_SELFSIM_SLACK", slack)
    sim, img = ctx.selfsim(frames, None, norms, af_start, 8, d, ext, 0.5)
    assert np.array_equal(img, want_img)
    np.testing.assert_allclose(sim, want_sim, rtol=0, atol=1e-5)
    _, img_only = ctx.selfsim(frames, None, norms, af_start, 8, d, ext, 0.5, want_sim=False)
    assert np.array_equal(img_only, want_img)


@pytest.mark.parametrize("knob,corr_frames,num_frames,decim", [("force", 6, 300, 1), ("force,5", 6, 300, 1),
                                                              ("force", 700, 2300, 16)])
def test_selfsim_pixels_that_need_the_sequential_order(ctx, monkeypatch, knob, corr_frames, num_frames, decim):
    """A quotient too close to an fp32 rounding boundary is evaluated in the reference's order of additions by a
    kernel of its own (through LDS; from global memory when the window is larger than LDS, here 700 frames; by
    the first kernel itself when their list is full, here 5 entries).  Rare on real data, so forced here for every
    pixel (SGZ_SELFSIM_SEQ, with error bounds widened until most pixels count as uncertain, SGZ_SELFSIM_SLACK, and a
    fix list of 3 entries that they overflow, so that every pixel is looked at)."""
    frames = _frames(num_frames)
    norms = norms_default()
    cfg = O.SelfSimConfig(0, 0, 0, 0, corr_frames * STEP, decim, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    ext, d, af_start, _ = O.selfsim_geometry(cfg, frames.shape[0], frames.shape[0])
    want_sim, want_img = O.selfsim_run(cfg, frames, None, norms)
    monkeypatch.setenv("SGZ_SELFSIM_FIX_CAP", "3")
    monkeypatch.setenv("SGZ_SELFSIM_SLACK", "3e-3")
    monkeypatch.setenv("SGZ_SELFSIM_SEQ", knob)
    sim, img = ctx.selfsim(frames, None, norms, af_start, corr_frames, d, ext, 0.5)
    assert ext > 50 and np.array_equal(img, want_img)
    assert np.array_equal(sim, want_sim)


def test_cross_file_similarity(ctx):
    """metaInput2: left windows from file 1, right windows from file 2 (impl/SelfSimilarityImpl.scala:131-141)."""
    f1, f2 = _frames(200, 3), _frames(180, 9)
    norms = norms_default()
    cfg = O.SelfSimConfig(0, 0, 0, 0, 6 * STEP, 2, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    ext, d, af_start, _ = O.selfsim_geometry(cfg, 200, 180)
    want_sim, _ = O.selfsim_run(cfg, f1, f2, norms, want_img=False)
    sim, _ = ctx.selfsim(f1, f2, norms, af_start, 6, d, ext, 0.5, want_img=False)
    # the reference mirrors the upper triangle (left <= right) into the lower one
    np.testing.assert_allclose(sim, want_sim, rtol=0, atol=1e-5)


def test_selfsim_matrix_for_colour_schemes_the_jvm_renders(ctx):
    """colors = psycho (the reference's default, SelfSimilarity.scala:178): the palette is a JVM-side class, so the
    route is the similarity matrix without a raster -- accepted for any colour id, within 1e-5 of the oracle --
    while asking the library for a psycho raster fails loudly.  (Parity of the palette itself: unpinned, third party.)"""
    import strugatzki_amd as S
    frames = _frames(260)
    norms = norms_default()
    cfg = O.SelfSimConfig(0, 0, 0, 0, 8 * STEP, 2, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    ext, d, af_start, _ = O.selfsim_geometry(cfg, frames.shape[0], frames.shape[0])
    want_sim, _ = O.selfsim_run(cfg, frames, None, norms, want_img=False)
    sim, img = ctx.selfsim(frames, None, norms, af_start, 8, d, ext, 0.5, colors=1, want_img=False)
    assert img is None
    np.testing.assert_allclose(sim, want_sim, rtol=0, atol=1e-5)
    with pytest.raises(S.NativeError):
        ctx.selfsim(frames, None, norms, af_start, 8, d, ext, 0.5, colors=1, want_img=True)


@pytest.mark.parametrize("two_files", [False, True])
def test_selfsim_rows_put_together_are_the_matrix(ctx, two_files):
    """sgz_selfsim_rows: bands of rows of the similarity matrix, for a caller that colours the similarities itself
    and cannot hold the whole matrix in one buffer.  The bands concatenated are the matrix of sgz_selfsim_run bit
    for bit; other inputs are not served from the matrix the context kept."""
    f1 = _frames(700, 3)
    f2 = _frames(650, 5) if two_files else None
    norms = norms_default()
    n2 = 650 if two_files else 700
    cfg = O.SelfSimConfig(0, 0, 0, 0, 6 * STEP, 1, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
    ext, d, af_start, _ = O.selfsim_geometry(cfg, 700, n2)
    full, _ = ctx.selfsim(f1, f2, norms, af_start, 6, d, ext, 0.5, want_img=False)
    cuts = [0, 1, 130, 131, 400, ext]
    bands = [ctx.selfsim_rows(f1, f2, norms, af_start, 6, d, ext, 0.5, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(bands, axis=0), full)
    assert ctx.selfsim_rows(f1, f2, norms, af_start, 6, d, ext, 0.5, 7, 7).shape == (0, ext)
    # another weight: a different matrix, not the one kept on the device
    other, _ = ctx.selfsim(f1, f2, norms, af_start, 6, d, ext, 0.25, want_img=False)
    got = ctx.selfsim_rows(f1, f2, norms, af_start, 6, d, ext, 0.25, 100, 140)
    assert np.array_equal(got, other[100:140]) and not np.array_equal(got, full[100:140])
    # other frames at the same address and of the same shape
    f1b = f1.copy()
    f1b[300:320] *= np.float32(1.5)
    other, _ = ctx.selfsim(f1b, f2, norms, af_start, 6, d, ext, 0.25, want_img=False)
    assert np.array_equal(ctx.selfsim_rows(f1b, f2, norms, af_start, 6, d, ext, 0.25, 280, 330), other[280:330])
    ctx.selfsim_release()
    assert np.array_equal(ctx.selfsim_rows(f1, f2, norms, af_start, 6, d, ext, 0.5, ext - 10, ext), full[ext - 10:])
    ctx.selfsim_release()
    import strugatzki_amd as S
    with pytest.raises(S.NativeError):
        ctx.selfsim_rows(f1, f2, norms, af_start, 6, d, ext, 0.5, 5, ext + 1)


def test_selfsim_rejects_bad_arguments(ctx):
    import strugatzki_amd as S
    frames = _frames(100)
    with pytest.raises(S.NativeError):
        ctx.selfsim(frames, None, None, 0, 4, 1, 10, 0.5, color_warp=0.0)
    with pytest.raises(S.NativeError):
        ctx.selfsim(frames, None, None, 0, 4, 1, 200, 0.5)  # windows past the end
    with pytest.raises(S.NativeError):
        ctx.selfsim(frames, None, None, 0, 4, 1, 10, 0.5, colors=1)  # psycho-optical palette not implemented


@pytest.mark.parametrize("fft,overlap", [(4096, 2), (8192, 4), (16, 1)])
def test_extraction_with_small_and_large_fft_sizes(ctx, fft, overlap):
    """The number of frames a workgroup takes follows the FFT size (its LDS share): the largest and the smallest
    size against the CPU statement."""
    rng = np.random.default_rng(fft)
    n = 6 * fft + 37
    t = np.arange(n, dtype=np.float32)
    pcm = (0.3 * np.sin(t * np.float32(2 * np.pi * 440.0 / 44100.0)) + 0.02 * rng.standard_normal(n)).astype(np.float32)
    want = O.extract(pcm, 44100.0, fft, overlap, 13, 0)
    got = ctx.extract(pcm, 44100.0, fft, overlap, 13, 0)
    assert got.shape == want.shape and got.shape[0] > 0
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
