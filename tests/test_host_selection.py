"""CPU tests of the library's host-side selection code (no GPU involved).

The selectors in libstrugatzki_amd.so (sgz_sel_*, sgz_segm_select) are fed the
oracle's dense similarities and must return exactly the oracle's matches /
breaks: same files, spans, sims and order
(impl/FeatureCorrelationImpl.scala:113-150,234,399-400;
impl/FeatureSegmentationImpl.scala:55-83,121-125).
"""
import numpy as np
import pytest

import strugatzki_amd as S
from strugatzki_amd import _native as N
from strugatzki_amd.engine import Selector, iter_files, match_tuple, segm_select
from common import NUM_CH, SEED, STEP, O, corr_config, make_plants, norms_default, synth_db, synth_target


def _oracle_case(num_matches, num_per_file, min_spacing, nf=10, frames=700, L=64, nplants=4, seed_shift=0):
    ts = 11
    plants = make_plants(nplants, L, nf, frames, ts)
    db = synth_db(nf, frames, plants, seed=SEED + seed_shift)
    tgt = synth_target(ts, L, seed=SEED + seed_shift)
    cfg = corr_config(ts, L, num_matches=num_matches, num_per_file=num_per_file, min_spacing=min_spacing)
    want, sims, boosts = O.correlation_run(cfg, tgt, norms_default(), db)
    return want, sims, boosts, L


CASES = [(10, 1, 0), (1, 1, 0), (5, 3, 0), (6, 2, 40 * STEP), (12, 4, 3 * STEP), (3, 5, 0), (8, 2, 700 * STEP),
         (4, 1, 10 ** 9)]


@pytest.mark.parametrize("num_matches,num_per_file,min_spacing", CASES)
def test_dense_replay_equals_oracle(num_matches, num_per_file, min_spacing):
    want, sims, boosts, L = _oracle_case(num_matches, num_per_file, min_spacing)
    sel = Selector(num_matches, num_per_file, min_spacing, STEP, L)
    for f, (s, b) in enumerate(zip(sims, boosts)):
        sel.feed_dense(f, s, b)
    got = sel.result()
    assert [match_tuple(g) for g in got] == [O.match_tuple(w) for w in want]


@pytest.mark.parametrize("num_matches,num_per_file,min_spacing", CASES)
def test_head_plus_candidates_replay_equals_oracle(num_matches, num_per_file, min_spacing):
    """Dense feeding until the set is full, then only windows above the threshold: same result."""
    want, sims, boosts, L = _oracle_case(num_matches, num_per_file, min_spacing, nf=14, seed_shift=3)
    sel = Selector(num_matches, num_per_file, min_spacing, STEP, L)
    f = 0
    while f < len(sims) and not sel.is_full:
        sel.feed_dense(f, sims[f], boosts[f])
        f += 1
    thr = sel.threshold
    for g in range(f, len(sims)):
        idx = np.flatnonzero(sims[g] > np.float32(thr)).astype(np.int32)
        sel.feed_candidates(g, idx, sims[g][idx], boosts[g][idx])
    assert [match_tuple(m) for m in sel.result()] == [O.match_tuple(w) for w in want]


PUNCH_CASES = [  # num_matches, num_per_file, min_spacing, min_punch, max_punch (feature frames)
    (5, 1, 0, 20, 60), (3, 2, 30 * STEP, 10, 25), (8, 3, 0, 0, 5), (1, 1, 0, 40, 40), (6, 2, 0, 30, 400),
    (4, 1, 0, 5, 3),  # maxPunch < minPunch: no pair at all
]


@pytest.mark.parametrize("num_matches,num_per_file,min_spacing,min_punch,max_punch", PUNCH_CASES)
def test_punch_out_replay_equals_oracle(num_matches, num_per_file, min_spacing, min_punch, max_punch):
    """sgz_sel_feed_file_punch over the oracle's dense punch-in / punch-out similarities."""
    from common import punch_config
    nf, frames, Lin, Lout, ts = 7, 420, 48, 32, 9
    db = synth_db(nf, [frames, 300, frames, 90, frames, 200, frames], make_plants(3, Lin + 60, nf, 200, ts))
    tgt = synth_target(ts, Lin + 120)
    cfg = punch_config(ts, Lin, ts + Lin + 20, Lout, min_punch=min_punch, max_punch=max_punch,
                       num_matches=num_matches, num_per_file=num_per_file, min_spacing=min_spacing)
    want, din, dout = O.correlation_run_punch(cfg, tgt, norms_default(), db)
    sel = Selector(num_matches, num_per_file, min_spacing, STEP, Lin)
    for f in range(nf):
        n = db[f].shape[0]  # (90 frames: shorter than the punch-in span, one window over the ring's stale contents)
        sel.feed_file_punch(f, n, Lin, Lout, min_punch, max_punch, din[f][0], din[f][1], dout[f][0], dout[f][1])
    got = sel.result()
    # (a negative inSim * outSim gives a NaN similarity, which the reference's gate lets in while the set has
    # room: compare through repr so that NaN equals NaN)
    assert repr([match_tuple(g) for g in got]) == repr([O.match_tuple(w) for w in want])
    if max_punch >= min_punch:
        assert len(got) > 0 and all(g.stop - g.start >= min_punch * STEP for g in got)
        assert all(g.stop - g.start <= max_punch * STEP for g in got)


def test_equal_sims_are_one_set_element():
    """Two windows gated to sim 0f: the set keeps the first, the global set does not grow twice."""
    L = 8
    sel = Selector(3, 2, 0, STEP, L)
    sims = np.zeros(40, dtype=np.float32)
    boosts = np.full(40, 9.0, dtype=np.float32)
    sel.feed_dense(0, sims, boosts)
    sel.feed_dense(1, sims, boosts)
    got = sel.result()
    assert len(got) == 1 and (got[0].file, got[0].start) == (0, 0)
    assert not sel.is_full


def test_overlapping_windows_collapse_with_min_spacing_zero():
    """spacing() is negative for overlapping windows, so minSpacing = 0 still collapses them (impl/SpanUtil.scala:38-43)."""
    L = 10
    sims = np.array([0.1, 0.5, 0.4, 0.9, 0.2] + [0.0] * 20 + [0.8], dtype=np.float32)
    boosts = np.ones_like(sims)
    sel = Selector(5, 5, 0, STEP, L)
    sel.feed_dense(0, sims, boosts)
    got = sel.result()
    # windows 0..4 overlap pairwise: they collapse onto the best (offset 3); offset 25 does not overlap offset 4
    assert [(m.start // STEP, m.sim) for m in got] == [(3, np.float32(0.9)), (25, np.float32(0.8))]


def test_nan_similarity_is_never_offered_once_full():
    L = 4
    sel = Selector(1, 1, 0, STEP, L)
    sel.feed_dense(0, np.array([0.3, 0.2], np.float32), np.ones(2, np.float32))
    assert sel.is_full
    sel.feed_dense(1, np.array([np.nan, 0.1], np.float32), np.ones(2, np.float32))
    got = sel.result()
    assert len(got) == 1 and got[0].file == 0 and got[0].sim == np.float32(0.3)


def test_iter_files_groups_sorted_candidates():
    c = np.zeros(5, dtype=N.CANDIDATE_DTYPE)
    c["file"] = [2, 2, 5, 7, 7]
    c["offset"] = [1, 9, 3, 0, 4]
    groups = [(g, list(part["offset"])) for g, part in iter_files(c)]
    assert groups == [(2, [1, 9]), (5, [3]), (7, [0, 4])]
    assert list(iter_files(c[:0])) == []


@pytest.mark.parametrize("num_breaks,min_spacing", [(1, 22050), (4, 22050), (6, 0), (3, 10 ** 7), (20, 5 * STEP)])
def test_break_selection_equals_oracle(num_breaks, min_spacing):
    frames = np.concatenate([O.synth_file(SEED, k, 300, NUM_CH) * np.float32(1.0 - 0.2 * (k % 2)) for k in range(4)])
    cfg = O.SegmConfig(1, 1, 50 * STEP, 1100 * STEP, 22050, 0.5, 1, num_breaks, min_spacing, STEP, NUM_CH - 1)
    want, dense = O.segmentation_run(cfg, frames, norms_default())
    got = segm_select(dense, 50, 43, STEP, num_breaks, min_spacing)
    assert [(b.sim, b.pos) for b in got] == [(w.sim, w.pos) for w in want]


def test_selector_rejects_bad_arguments():
    with pytest.raises(ValueError):
        Selector(-1, 1, 0, STEP, 10)
    with pytest.raises(ValueError):
        Selector(1, 1, 0, 0, 10)


def _ring_by_simulation(counts, L):
    """The reference's ring, written frame by frame (impl/FeatureCorrelationImpl.scala:186-195, 242-245): the first
    round of a file reads min(frames, L) frames to position 0, every later frame goes to readOff, which advances
    modulo L.  Returns, per file shorter than L, what positions [own, L) hold when the file is scored."""
    ring = [(-1, 0)] * L
    out = {}
    for f, n in enumerate(counts):
        if n <= 0:
            continue
        first = min(n, L)
        for p in range(first):
            ring[p] = (f, p)
        if n < L:
            out[f] = list(ring[n:])
        read_off, frame = first % L, first
        while frame < n:
            ring[read_off] = (f, frame)
            read_off = (read_off + 1) % L
            frame += 1
    return out


@pytest.mark.parametrize("seed", range(8))
def test_ring_runs_equal_a_frame_by_frame_simulation_of_the_reference_ring(seed):
    from strugatzki_amd.engine import ring_runs
    rng = np.random.default_rng(seed)
    L = int(rng.integers(1, 40))
    counts = [int(rng.choice([0, 1, L - 1, L, L + 1, 2 * L, 3 * L + 2, int(rng.integers(0, 5 * L + 3))]))
              for _ in range(int(rng.integers(1, 14)))]
    counts = [max(c, 0) for c in counts]
    want = _ring_by_simulation(counts, L)
    runs = ring_runs(counts, L)
    got = {}
    for r in runs:
        cells = got.setdefault(int(r["file"]), {})
        for k in range(int(r["count"])):
            src = int(r["src_file"])
            cells[int(r["dest"]) + k] = (src, int(r["src_frame"]) + k if src >= 0 else 0)
    assert set(got) == set(want)
    for f, cells in want.items():
        own = counts[f]
        assert [got[f][own + k] for k in range(L - own)] == cells
