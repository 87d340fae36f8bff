"""Time of a search whose target is new (same length): sgz_corr_set_target + the search, against the search alone
with the target already set -- what a query service pays per query over a resident database.
usage: new_target_timing.py [searches] [files]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import strugatzki_amd as S
from common import STEP, norms_default

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 120
L, FR = 2584, 25839
rng = np.random.default_rng(3)
with S.Context(0) as ctx:
    ctx.db_reset(14)
    for f in range(nf):
        ctx.db_add_synthetic(0x5354525547, f, FR)
    ctx.set_norms(norms_default())
    src = ctx.db_read_file(0)
    tgts = []
    for i in range(16):
        o = int(rng.integers(0, FR - L))
        t = src[o:o + L].copy() + (rng.standard_normal((L, 14)) * 0.05).astype(np.float32)
        t[:, 0] = np.abs(t[:, 0]) + np.float32(0.01)
        tgts.append(np.ascontiguousarray(t))
    p = S.SearchParams(10, 1, 0, STEP)
    for t in tgts:  # warm
        ctx.corr_set_target(t, 0.5, 8.0)
        S.search(ctx, p)
    t0 = time.perf_counter()
    for i in range(n):
        ctx.corr_set_target(tgts[i % 16], 0.5, 8.0)
        S.search(ctx, p)
    t1 = time.perf_counter()
    for i in range(n):
        S.search(ctx, p)
    t2 = time.perf_counter()
    # the two calls of a query apart (host time of each: the target call returns when the device has reported the
    # target's statistics, the search when its matches are there)
    a = b = 0.0
    for i in range(n):
        q0 = time.perf_counter()
        ctx.corr_set_target(tgts[i % 16], 0.5, 8.0)
        q1 = time.perf_counter()
        S.search(ctx, p)
        q2 = time.perf_counter()
        a += q1 - q0
        b += q2 - q1
    print(f"in a query: set_target {1e3 * a / n:.4f} ms, search {1e3 * b / n:.4f} ms", flush=True)
    t3 = time.perf_counter()
    for i in range(n):
        ctx.corr_set_target(tgts[i % 16], 0.5, 8.0)
    ctx.synchronize()
    t4 = time.perf_counter()
    print(f"new target + search {1e3 * (t1 - t0) / n:.4f} ms, search alone {1e3 * (t2 - t1) / n:.4f} ms, "
          f"set_target alone {1e3 * (t4 - t3) / n:.4f} ms", flush=True)
