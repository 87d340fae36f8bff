"""Several queries in flight on one GPU (sgz_corr_search_batch): every query's result is the oracle's result for
its target -- offsets bit-exact, similarities within 1e-5 -- and the very records a single context returns, whatever
context a query lands on and in whatever order the queries finish."""
import numpy as np
import pytest

from common import NUM_CH, STEP, HipMem, O, corr_config, make_plants, norms_default, synth_db, synth_target

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _load(ctx, db, norms):
    ctx.db_reset(NUM_CH)
    for f in db:
        ctx.db_add_file(f)
    ctx.set_norms(norms)


def _same(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert (g.file, g.start, g.stop) == (w.file, w.start, w.stop)
        assert abs(g.sim - w.sim) <= TOL
        assert abs(g.boost_in - w.boost_in) <= 2e-6 * abs(w.boost_in)


@pytest.mark.parametrize("L,frames,lanes,on_device", [
    (96, [900] * 10, 3, False),                                  # matrix-core bounds
    (1100, [5000, 3000, 4100, 700, 5000, 2222, 3500], 3, False),  # spectral bounds, ragged files, one shorter than the target
    (1100, [5000, 3000, 4100, 700, 5000, 2222, 3500], 2, True),
    (96, [900] * 10, 4, True),
])
def test_batch_equals_single_queries_and_oracle(L, frames, lanes, on_device):
    import strugatzki_amd as S
    nf = len(frames)
    ts0 = 17
    plants = make_plants(3, L, nf, min(fr for fr in frames if fr > L), ts0)  # (in files 0, 2 * (nf // 3), ...: long ones)
    assert all(frames[pl["file"]] >= pl["offset"] + L for pl in plants)
    db = synth_db(nf, frames, plants)
    norms = norms_default()
    starts = [ts0, 60, 333, 128, 17, 501, 250, 90, 410]  # (target 0 twice: the same query on two contexts)
    tfile = synth_target(max(starts), L)
    p = S.SearchParams(5, 1, 0, STEP, capacity=1 << 16)
    ctxs = [S.Context(0) for _ in range(lanes)]
    mem = HipMem()
    try:
        for c in ctxs:
            _load(c, db, norms)
        targets = [np.ascontiguousarray(tfile[s:s + L]) for s in starts]
        if on_device:
            handles = []
            for t in targets:
                d = mem.alloc(t.nbytes)
                mem.upload(d, t)
                handles.append(d)
            got = S.search_batch(ctxs, p, handles, 0.5, 8.0, target_frames=L)
        else:
            got = S.search_batch(ctxs, p, targets, 0.5, 8.0)
        assert len(got) == len(starts)
        for q, s in enumerate(starts):
            want, _, _ = O.correlation_run(corr_config(s, L, num_matches=5), tfile, norms, db, dense=False)
            _same(got[q], want)
            ctxs[0].corr_set_target(targets[q], 0.5, 8.0)
            one = S.search(ctxs[0], p)
            assert [(m.file, m.start, m.stop, m.sim, m.boost_in) for m in one] == \
                   [(m.file, m.start, m.stop, m.sim, m.boost_in) for m in got[q]]
        # more contexts than queries, and an empty batch
        got1 = S.search_batch(ctxs, p, targets[:1], 0.5, 8.0)
        assert [(m.file, m.start, m.sim) for m in got1[0]] == [(m.file, m.start, m.sim) for m in got[0]]
        assert S.search_batch(ctxs, p, [], 0.5, 8.0) == []
    finally:
        mem.free_all()
        for c in ctxs:
            c.close()


def test_batch_refuses_the_same_context_twice():
    import strugatzki_amd as S
    c = S.Context(0)
    try:
        _load(c, synth_db(2, 600), norms_default())
        t = synth_target(10, 64)[10:74]
        with pytest.raises(S.NativeError):
            S.search_batch([c, c], S.SearchParams(3, 1, 0, STEP), [t, t], 0.5, 8.0)
    finally:
        c.close()


def test_batch_reports_a_failing_query():
    import strugatzki_amd as S
    db = synth_db(3, 600)
    ctxs = [S.Context(0) for _ in range(2)]
    try:
        _load(ctxs[0], db, norms_default())  # the second context holds no database
        t = synth_target(10, 64)[10:74]
        with pytest.raises(S.NativeError):
            S.search_batch(ctxs, S.SearchParams(3, 1, 0, STEP), [t, t, t, t], 0.5, 8.0)
    finally:
        for c in ctxs:
            c.close()
