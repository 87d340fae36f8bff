"""sgz_corr_search_multi with the candidate exchange over RCCL, in a process of its own that never loads torch
(the library takes RCCL from the ROCm installation of its own HIP runtime; a process that also holds the copies a
Python package bundles could not tell which one a failure belongs to).  Started by test_gpu_multi.py.
Prints one JSON line: per case the exchange that was used, whether the result equals the oracle's, and the reason
the library gave when it merged in host memory instead."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from common import NUM_CH, STEP, O, corr_config, make_plants, norms_default, synth_db, synth_target  # noqa: E402

import strugatzki_amd as S  # noqa: E402
from strugatzki_amd import _native as N  # noqa: E402


def run(num_ctx, frames, L, ts, nm, plants_n, constant=None):
    nf = len(frames)
    plants = [q for q in make_plants(plants_n, L, nf, max(frames), ts) if q["offset"] + L <= frames[q["file"]]]
    db = synth_db(nf, frames, plants)
    if constant is not None:
        f, a, b = constant
        db[f][a:b] = db[f][a]
    tgt = synth_target(ts, L)
    norms = norms_default()
    want, _, _ = O.correlation_run(corr_config(ts, L, num_matches=nm), tgt, norms, db, dense=False)
    ctxs, ids = [], []
    for c in range(num_ctx):
        ctx = S.Context(0)
        mine = [g for g in range(nf) if g % num_ctx == c]
        ctx.db_reset(NUM_CH)
        for g in mine:
            ctx.db_add_file(db[g])
        ctx.set_norms(norms)
        ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
        ctxs.append(ctx)
        ids.append(mine)
    try:
        assert N.lib.sgz_multi_set_exchange(ctxs[0]._h, 1) == 0
        out = []
        for _ in range(2):  # the second search reuses communicators and buffers
            got = S.search_multi(ctxs, ids, nm, 1, 0, STEP)
            same = [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want] and \
                all(abs(g.sim - w.sim) <= 1.2e-7 or (g.sim != g.sim and w.sim != w.sim) for g, w in zip(got, want))
            out.append(dict(used=N.lib.sgz_multi_last_exchange(ctxs[0]._h), same=same, matches=len(got),
                            why=N.lib.sgz_last_error(ctxs[0]._h).decode()))
        return out
    finally:
        for c in ctxs:
            c.close()


def main():
    res = {
        # one context: one communicator rank, the all-gather carries its own block
        "one": run(1, [900] * 9, 96, 17, 4, 5),
        # files shorter than the target, an empty file, a constant stretch (NaN similarities: the bounds cannot decide)
        "one_awkward": run(1, [60, 1000, 119, 7, 120, 121, 0, 700, 1143, 30, 400, 1, 260], 120, 5, 5, 0, (7, 0, 200)),
        # two contexts on the one GPU of this box: RCCL wants distinct devices
        "two_on_one_gpu": run(2, [900] * 9, 96, 17, 4, 5),
    }
    print(json.dumps(res))
    return 0


if __name__ == "__main__":
    sys.exit(main())
