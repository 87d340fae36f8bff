"""sgz_corr_search_multi: a database spread over several contexts, searched from one process through the C ABI.
Bar: the list equals the oracle's over all files in global order -- offsets bit-exact, similarities within 1e-5 --
whatever the number of contexts and the way the files are dealt, files shorter than the target included (their
ring contents come from files other contexts hold)."""
import numpy as np
import pytest

from common import NUM_CH, STEP, O, corr_config, make_plants, norms_default, synth_db, synth_target

pytestmark = pytest.mark.gpu


def _contexts(S, n, db, deal, norms, target, weight=0.5):
    ctxs, ids = [], []
    for c in range(n):
        ctx = S.Context(0)
        mine = [g for g in range(len(db)) if deal(g) == c]
        ctx.db_reset(NUM_CH)
        for g in mine:
            ctx.db_add_file(db[g])
        ctx.set_norms(norms)
        ctx.corr_set_target(target, weight, 8.0)
        ctxs.append(ctx)
        ids.append(mine)
    return ctxs, ids


def _same(got, want, tol=1e-5):
    assert [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
    for g, w in zip(got, want):
        assert abs(g.sim - w.sim) <= tol or (g.sim != g.sim and w.sim != w.sim)
        assert abs(g.boost_in - w.boost_in) <= 2e-6 * abs(w.boost_in)


@pytest.mark.parametrize("num_ctx", [1, 2, 3])
@pytest.mark.parametrize("num_matches,num_per_file,min_spacing", [(5, 1, 0), (6, 2, 30 * STEP), (3, 1, 0)])
def test_multi_context_search_equals_the_oracle(num_ctx, num_matches, num_per_file, min_spacing, screen_kernel_env):
    import strugatzki_amd as S
    L, ts, nf, frames = 96, 17, 13, 900
    db = synth_db(nf, frames, make_plants(5, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    norms = norms_default()
    want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=num_matches, num_per_file=num_per_file,
                                               min_spacing=min_spacing), tgt, norms, db, dense=False)
    ctxs, ids = _contexts(S, num_ctx, db, lambda g: g % num_ctx, norms, tgt[ts:ts + L])
    try:
        got = S.search_multi(ctxs, ids, num_matches, num_per_file, min_spacing, STEP)
        _same(got, want, 1.2e-7 if num_per_file == 1 else 1e-5)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("num_ctx", [1, 2, 4])
def test_multi_context_search_with_files_shorter_than_the_target(num_ctx, screen_kernel_env):
    """Short files see the ring of the single loop over ALL files: their stale frames come from files that live
    in other contexts.  Also a file that starts with a constant stretch (NaN similarities) and an empty file."""
    import strugatzki_amd as S
    L, ts = 120, 5
    frames = [60, 1000, 119, 7, 120, 121, 0, 700, 1024 + 119, 30, 400, 1, 260]
    db = synth_db(len(frames), frames)
    db[7][0:200] = db[7][0]
    tgt = synth_target(ts, L)
    norms = norms_default()
    for nm, npf in ((4, 1), (8, 1), (5, 3)):
        want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=nm, num_per_file=npf), tgt, norms, db,
                                       dense=False)
        for deal in (lambda g: g % num_ctx, lambda g: (g * 7 // 3) % num_ctx):
            ctxs, ids = _contexts(S, num_ctx, db, deal, norms, tgt[ts:ts + L])
            try:
                got = S.search_multi(ctxs, ids, nm, npf, 0, STEP)
                _same(got, want)
            finally:
                for c in ctxs:
                    c.close()


def test_multi_context_search_with_the_exchange_over_rccl():
    """sgz_multi_set_exchange(1): the contexts' candidate blocks travel through one RCCL all-gather (a communicator
    rank per context) and the selection replays over the first context's copy.  This box has one GPU: one context
    exchanges over RCCL for real; two contexts on the same GPU are what RCCL refuses (one rank per device), and the
    library then merges in host memory and says why.  Every result equals the oracle's."""
    import json
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi_rccl_child.py")
    pr = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=150)
    assert pr.returncode == 0, pr.stderr[-3000:]
    res = json.loads(pr.stdout.strip().splitlines()[-1])
    for case in ("one", "one_awkward"):
        for r in res[case]:
            assert r["same"] and r["used"] == 1, (case, r)
    for r in res["two_on_one_gpu"]:
        assert r["same"] and r["used"] == 0 and "RCCL" in r["why"], r


def test_multi_context_search_rejects_inconsistent_input():
    import strugatzki_amd as S
    L, ts = 40, 3
    db = synth_db(4, 300)
    tgt = synth_target(ts, L)
    ctxs, ids = _contexts(S, 2, db, lambda g: g % 2, norms_default(), tgt[ts:ts + L])
    try:
        with pytest.raises(S.NativeError):
            S.search_multi(ctxs, [[0, 2], [1, 1]], 2, 1, 0, STEP)  # a file twice, one missing
        with pytest.raises(S.NativeError):
            S.search_multi(ctxs, [[2, 0], [1, 3]], 2, 1, 0, STEP)  # not ascending
        ctxs[1].corr_set_target(tgt[ts:ts + L - 1], 0.5, 8.0)
        with pytest.raises(S.NativeError):
            S.search_multi(ctxs, ids, 2, 1, 0, STEP)  # different targets
    finally:
        for c in ctxs:
            c.close()
