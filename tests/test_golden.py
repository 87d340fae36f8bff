"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU part: the oracle still reproduces them bit for bit.
GPU part: the HIP path reproduces them (sims within 1e-5, offsets / positions / pixels exact).
"""
import os

import numpy as np
import pytest

from common import NUM_CH, STEP, O, corr_config, punch_config

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


# ---------------------------------------------------------------- oracle vs golden (CPU)
def test_oracle_reproduces_correlation_golden():
    g = _load("correlation.npz")
    db = [np.ascontiguousarray(x) for x in g["db"]]
    L, ts = int(g["L"]), int(g["ts"])
    for k in range(3):
        nm, npf, ms = (int(v) for v in g[f"setting{k}"])
        cfg = corr_config(ts, L, num_matches=nm, num_per_file=npf, min_spacing=ms)
        m, sims, boosts = O.correlation_run(cfg, g["target"], g["norms"], db)
        assert np.array_equal(np.array([O.match_tuple(x) for x in m], dtype=np.float64), g[f"matches{k}"])
    assert np.array_equal(np.stack(sims), g["sims"]) and np.array_equal(np.stack(boosts), g["boosts"])
    cfg = corr_config(ts, L, weight=0.2, normalize=False, num_matches=3)
    _, sims, _ = O.correlation_run(cfg, g["target"], None, db)
    assert np.array_equal(np.stack(sims), g["sims_raw_w02"])


def _punch_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.PUNCH


def test_oracle_reproduces_punch_golden():
    g, q = _load("punch.npz"), _punch_cases()
    db = [np.ascontiguousarray(x) for x in g["db"]]
    for k, (nm, npf, ms, lo, hi) in enumerate(q["settings"]):
        cfg = punch_config(q["ts"], q["Lin"], q["ts"] + q["Lin"] + q["gap"], q["Lout"], min_punch=lo, max_punch=hi,
                           out_weight=0.25, num_matches=nm, num_per_file=npf, min_spacing=ms)
        m, din, dout = O.correlation_run_punch(cfg, g["target"], g["norms"], db)
        assert np.array_equal(np.array([O.match_tuple(x) for x in m], dtype=np.float64), g[f"matches{k}"],
                              equal_nan=True)
        assert np.array_equal(np.stack(din), g[f"dense_in{k}"], equal_nan=True)
        assert np.array_equal(np.stack(dout), g[f"dense_out{k}"], equal_nan=True)


@pytest.mark.gpu
def test_gpu_reproduces_punch_golden(ctx):
    import strugatzki_amd as S
    g, q = _load("punch.npz"), _punch_cases()
    ctx.db_reset(NUM_CH)
    for x in g["db"]:
        ctx.db_add_file(np.ascontiguousarray(x))
    ctx.set_norms(g["norms"])
    tgt, ts, o0 = g["target"], q["ts"], q["ts"] + q["Lin"] + q["gap"]
    for k, (nm, npf, ms, lo, hi) in enumerate(q["settings"]):
        got = S.search_punch(ctx, S.SearchParams(nm, npf, ms, STEP), tgt[ts:ts + q["Lin"]], 0.5,
                             tgt[o0:o0 + q["Lout"]], 0.25, 8.0, lo, hi)
        want = g[f"matches{k}"]
        assert len(got) == len(want) > 0
        for m, w in zip(got, want):
            assert (m.file, m.start, m.stop) == (int(w[1]), int(w[2]), int(w[3]))
            assert abs(m.sim - w[0]) <= 1e-5 or (np.isnan(m.sim) and np.isnan(w[0]))


def test_oracle_reproduces_segmentation_golden():
    g = _load("segmentation.npz")
    a0, a1, H, nb, ms = (int(v) for v in g["params"])
    cfg = O.SegmConfig(1, 1, a0 * STEP, a1 * STEP, 22050, 0.5, 1, nb, ms, STEP, NUM_CH - 1)
    breaks, dense = O.segmentation_run(cfg, g["frames"], g["norms"])
    assert np.array_equal(dense, g["dense"])
    assert np.array_equal(np.array([(b.sim, b.pos) for b in breaks], dtype=np.float64), g["breaks"])


def test_oracle_reproduces_selfsim_golden():
    g = _load("selfsim.npz")
    cfg = O.SelfSimConfig(0, 0, 0, 0, 8 * STEP, 3, 0.5, 0, 0.7, 0.9, 0, 1, STEP, NUM_CH - 1)
    sim, img = O.selfsim_run(cfg, g["frames"], None, g["norms"])
    assert np.array_equal(sim, g["sim"]) and np.array_equal(img, g["img"])


# ---------------------------------------------------------------- HIP vs golden (GPU)
@pytest.mark.gpu
def test_gpu_reproduces_correlation_golden(ctx):
    import strugatzki_amd as S
    g = _load("correlation.npz")
    L, ts = int(g["L"]), int(g["ts"])
    ctx.db_reset(NUM_CH)
    for x in g["db"]:
        ctx.db_add_file(np.ascontiguousarray(x))
    ctx.set_norms(g["norms"])
    ctx.corr_set_target(g["target"][ts:ts + L], 0.5, 8.0)
    ctx.corr_run()
    for f in range(len(g["db"])):
        sim, boost = ctx.corr_read_dense(f)
        np.testing.assert_allclose(sim, g["sims"][f], rtol=0, atol=1e-5)
        np.testing.assert_allclose(boost, g["boosts"][f], rtol=2e-6, atol=0)
    for k in range(3):
        nm, npf, ms = (int(v) for v in g[f"setting{k}"])
        got = S.search(ctx, S.SearchParams(nm, npf, ms, STEP))
        want = g[f"matches{k}"]
        assert len(got) == len(want)
        for m, w in zip(got, want):
            assert (m.file, m.start, m.stop) == (int(w[1]), int(w[2]), int(w[3]))
            assert abs(m.sim - w[0]) <= 1e-5
    ctx.set_norms(None)
    ctx.corr_set_target(g["target"][ts:ts + L], 0.2, 8.0)
    ctx.corr_run()
    for f in range(len(g["db"])):
        sim, _ = ctx.corr_read_dense(f)
        np.testing.assert_allclose(sim, g["sims_raw_w02"][f], rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_gpu_reproduces_segmentation_golden(ctx):
    from strugatzki_amd.engine import segm_select
    g = _load("segmentation.npz")
    a0, a1, H, nb, ms = (int(v) for v in g["params"])
    dense = ctx.segm_curve(g["frames"], g["norms"], a0, a1, H, 0.5)
    assert np.array_equal(dense, g["dense"])  # same summation order as the reference: bit-exact
    got = segm_select(dense, a0, H, STEP, nb, ms)
    assert np.array_equal(np.array([(b.sim, b.pos) for b in got], dtype=np.float64), g["breaks"])


@pytest.mark.gpu
def test_gpu_reproduces_selfsim_golden(ctx):
    g = _load("selfsim.npz")
    ext, decim, af_start, num_corrs, H = (int(v) for v in g["params"])
    sim, img = ctx.selfsim(g["frames"], None, g["norms"], af_start, H, decim, ext, 0.5, 0, 0.7, 0.9, False)
    np.testing.assert_allclose(sim, g["sim"], rtol=0, atol=1e-5)
    # gray levels are integer work: every pixel whose level the similarity's error could change is evaluated in
    # the reference's own order, so the raster is the oracle's
    assert np.array_equal(img, g["img"])
