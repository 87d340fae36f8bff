"""Many searches in a row over one resident database with changing targets (same length: everything cached stays),
each compared with the search at reference arithmetic (fp64 kernel over every window): exercises the state that
alternates between searches (per-file bound arrays, candidate and list counters, implicit boosts).
usage: soak_search.py [searches] [files] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import strugatzki_amd as S
from common import STEP, norms_default

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
L, FR = 2584, 25839
rng = np.random.default_rng(seed)
bad = 0
with S.Context(0) as ctx:
    ctx.db_reset(14)
    for f in range(nf):
        ctx.db_add_synthetic(0x5354525547, f, FR)
    ctx.set_norms(norms_default())
    src = [ctx.db_read_file(f) for f in range(min(nf, 4))]
    for i in range(n):
        f = int(rng.integers(0, len(src)))
        o = int(rng.integers(0, FR - L))
        tgt = src[f][o:o + L].copy()
        tgt += (rng.standard_normal(tgt.shape) * rng.choice([0.0, 0.01, 0.1])).astype(np.float32)
        tgt[:, 0] = np.abs(tgt[:, 0]) + np.float32(0.01)
        k = int(rng.integers(1, 12))
        ctx.corr_set_target(tgt, float(rng.choice([0.0, 0.5, 1.0])), float(rng.choice([2.0, 8.0])))
        got = S.search(ctx, S.SearchParams(k, 1, 0, STEP))
        if i % 10 == 0:  # (the exact search costs forty times the screened one)
            want = S.search(ctx, S.SearchParams(k, 1, 0, STEP, screen=False))
            same = [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want] and all(
                abs(g.sim - w.sim) <= 1e-5 for g, w in zip(got, want))
            if not same:
                bad += 1
                print("MISMATCH at search", i, [(g.file, g.start, g.sim) for g in got][:3], [(w.file, w.start, w.sim) for w in want][:3], flush=True)
        if i % 200 == 199:
            print(i + 1, "searches,", bad, "mismatches", flush=True)
print(n, "searches,", bad, "mismatches")
sys.exit(1 if bad else 0)
