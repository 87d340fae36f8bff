"""Random FeatureSegmentation, SelfSimilarity, FeatureStats, CrossSimilarity and FeatureExtraction inputs on the GPU against the
oracle (ragged sizes, channel counts, awkward data).  usage: fuzz_halfcorr.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import SEED, STEP, O, norms_default
import strugatzki_amd as S
from strugatzki_amd.engine import segm_select


def _data(rng, n, num_ch):
    x = O.synth_file(SEED, int(rng.integers(0, 1000)), n, num_ch)
    kind = int(rng.integers(0, 8))
    if kind == 0 and n > 12:
        a = int(rng.integers(0, n - 6))
        x[a:a + int(rng.integers(3, 200))] = x[a]
    elif kind == 1:
        x *= np.float32(rng.choice([0.01, 30.0, 300.0]))
    elif kind == 2:
        x[:, 1:] += np.linspace(-3.0, 5.0, num_ch - 1, dtype=np.float32)
    elif kind == 3:
        x[n // 2:] += np.float32(2.5)
    elif kind == 4:
        x[:] = x[0]
    return x


def _same(a, b, atol):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.all(np.abs(a[~na] - b[~nb]) <= atol))


def _segm(ctx, rng):
    num_ch = int(rng.choice([2, 5, 14, 14, 21, 64]))
    n = int(rng.integers(4, 1600))
    frames = _data(rng, n, num_ch)
    normalize = bool(rng.random() < 0.8)
    norms = norms_default(num_ch) if normalize else None
    H = int(rng.integers(1, max(2, min(n // 2, 90))))
    corr_len = H * STEP + int(rng.integers(0, STEP // 2))
    H = O.full_to_feat(corr_len, STEP)
    s0 = int(rng.integers(0, n // 2)) * STEP if rng.random() < 0.4 else None
    s1 = int(rng.integers(n // 2, n + 20)) * STEP if rng.random() < 0.4 else None
    weight = float(rng.choice([0.0, 0.3, 0.5, 1.0]))
    nb, sp = int(rng.integers(1, 9)), int(rng.choice([0, 10 * STEP, 22050]))
    cfg = O.SegmConfig(int(s0 is not None), int(s1 is not None), s0 or 0, s1 or 0, corr_len, weight, int(normalize), nb,
                       sp, STEP, num_ch - 1)
    want, dense = O.segmentation_run(cfg, frames, norms)
    a0 = max(0, O.full_to_feat(s0, STEP)) if s0 is not None else 0
    a1 = min(n, O.full_to_feat(s1, STEP)) if s1 is not None else n
    got = ctx.segm_curve(frames, norms, a0, a1, H, weight)
    if not _same(got, dense, 1e-5):
        return f"segm curve ch={num_ch} n={n} H={H} span={s0},{s1} w={weight} norm={normalize}"
    br = segm_select(got, a0, H, STEP, nb, sp)
    if repr([(b.sim, b.pos) for b in br]) != repr([(w.sim, w.pos) for w in want]):
        if np.array_equal(got, dense, equal_nan=True):
            return f"segm breaks ch={num_ch} n={n} H={H} span={s0},{s1} nb={nb} sp={sp}"
    return None


def _selfsim(ctx, rng):
    num_ch = int(rng.choice([2, 5, 14, 14, 21, 64]))
    n1 = int(rng.integers(12, 420))
    two = rng.random() < 0.3
    n2 = int(rng.integers(12, 420)) if two else n1
    f1 = _data(rng, n1, num_ch)
    f2 = _data(rng, n2, num_ch) if two else None
    normalize = bool(rng.random() < 0.8)
    norms = norms_default(num_ch) if normalize else None
    H = int(rng.integers(1, max(2, min(n1, n2) // 3)))
    decim = int(rng.integers(1, 5))
    weight = float(rng.choice([0.0, 0.5, 1.0]))
    warp, ceil, inv = float(rng.choice([0.5, 1.0, 2.0])), float(rng.choice([0.7, 1.0])), bool(rng.integers(0, 2))
    cfg = O.SelfSimConfig(0, 0, 0, 0, H * STEP, decim, weight, 0, warp, ceil, int(inv), int(normalize), STEP, num_ch - 1)
    ext, d, af_start, _ = O.selfsim_geometry(cfg, n1, n2)
    if ext < 1:
        return None
    want_sim, want_img = O.selfsim_run(cfg, f1, f2, norms)
    sim, img = ctx.selfsim(f1, f2, norms, af_start, H, d, ext, weight, 0, warp, ceil, inv)
    if not _same(sim, want_sim, 1e-5):
        return f"selfsim matrix ch={num_ch} n={n1},{n2} H={H} decim={decim} w={weight} norm={normalize}"
    dpx = np.abs((img & 0xFF).astype(np.int64) - (want_img & 0xFF).astype(np.int64))
    if dpx.max() > 1:
        return f"selfsim image ch={num_ch} n={n1},{n2} H={H} decim={decim} warp={warp} ceil={ceil} inv={inv}"
    return None


def _stats(ctx, rng):
    num_ch = int(rng.choice([2, 5, 14, 21, 40]))
    db = [_data(rng, int(rng.integers(1, 3000)), num_ch) for _ in range(int(rng.integers(1, 6)))]
    want = O.feature_stats(db)
    ctx.db_reset(num_ch)
    for x in db:
        ctx.db_add_file(x)
    got = ctx.feature_stats()
    if not (np.array_equal(np.isnan(got), np.isnan(want)) and
            np.allclose(got[~np.isnan(got)], want[~np.isnan(want)], rtol=1e-12, atol=0)):
        return f"stats ch={num_ch} frames={[x.shape[0] for x in db]}\n  got {got.tolist()}\n  want {want.tolist()}"
    return None


def _cross(ctx, rng):
    """CrossSimilarity through the calls its processor makes (shorter span = target, longer = the one file)."""
    num_ch = int(rng.choice([2, 5, 14, 14, 21, 64]))
    na, nb = int(rng.integers(1, 500)), int(rng.integers(1, 2500))
    fa, fb = _data(rng, na, num_ch), _data(rng, nb, num_ch)
    fa[:, 0], fb[:, 0] = np.abs(fa[:, 0]) + np.float32(0.01), np.abs(fb[:, 0]) + np.float32(0.01)
    normalize = bool(rng.random() < 0.8)
    norms = norms_default(num_ch) if normalize else None
    weight, max_boost = float(rng.choice([0.0, 0.5, 1.0])), float(rng.choice([1.5, 8.0, 50.0]))
    if rng.random() < 0.5:
        fa, fb = fb, fa
    want = O.crosssim_run(fa, fb, norms, weight, max_boost)
    tg, db = (fa, fb) if fa.shape[0] < fb.shape[0] else (fb, fa)
    ctx.db_reset(num_ch)
    ctx.db_add_file(db)
    ctx.set_norms(norms)
    ctx.corr_set_target(tg, weight, max_boost)
    ctx.corr_run(0, 1)
    got = ctx.corr_read_dense_range(0, 1)[0][0]
    if not _same(got, want, 1e-5):
        return f"crosssim ch={num_ch} n={fa.shape[0]},{fb.shape[0]} w={weight} mb={max_boost} norm={normalize}"
    return None


def _extract(ctx, rng):
    """FeatureExtraction on random PCM: lengths around the framing's edges, channel behaviours, fft sizes."""
    fft = int(rng.choice([256, 512, 1024, 2048, 4096, 8192]))
    ov = int(rng.choice([1, 2, 4]))
    nco = int(rng.choice([5, 13, 20]))
    chans = int(rng.choice([1, 1, 2, 3]))
    n = int(rng.choice([1, fft // ov - 1, fft // ov, fft, fft + 1, int(rng.integers(2, 60000 + 8 * fft))]))
    sr = float(rng.choice([22050.0, 44100.0, 48000.0]))
    t = np.arange(n, dtype=np.float32)[:, None]
    f0 = rng.uniform(50.0, 4000.0, size=(1, chans)).astype(np.float32)
    pcm = (0.3 * np.sin(t * (2 * np.pi * f0 / sr)) + 0.05 * rng.standard_normal((n, chans))).astype(np.float32)
    kind = int(rng.integers(0, 6))
    if kind == 0:
        pcm[:] = 0.0
    elif kind == 1:
        pcm *= np.float32(1e-4)
    elif kind == 2 and n > 10:
        pcm[n // 3: n // 2] = 0.0
    behav = int(rng.integers(0, 3))
    want = O.extract(pcm, sr, fft, ov, nco, behav)
    got = ctx.extract(pcm, sr, fft, ov, nco, behav)
    if got.shape != want.shape or not _same(got, want, 2e-5):
        worst = float(np.nanmax(np.abs(got - want))) if got.shape == want.shape and got.size else -1.0
        return f"extract n={n} ch={chans} fft={fft} ov={ov} coeffs={nco} behav={behav} sr={sr} kind={kind} worst={worst}"
    return None


def run(cases: int, seed: int, ctx=None, verbose=True) -> int:
    rng = np.random.default_rng(seed)
    own = ctx is None
    if own:
        ctx = S.Context(0)
    bad = 0
    for case in range(cases):
        fn = (_segm, _selfsim, _stats, _cross, _extract)[case % 5]
        msg = fn(ctx, rng)
        if msg:
            bad += 1
            if verbose:
                print(f"MISMATCH case {case}: {msg}", flush=True)
    if own:
        ctx.close()
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n} cases, {bad} mismatches", flush=True)
    sys.exit(1 if bad else 0)
