"""Queries in flight at once on one GPU: K contexts, each with its own copy of the database and its own host
thread, every thread running queries (a new target, then the search) back to back.  A query alone is a chain of
seven dependent launches of which only the screening kernel fills the chip; with several in flight the short
kernels of one query run beside the screening kernel of another.
usage: concurrent_queries.py [queries per thread] [files] [max contexts]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import strugatzki_amd as S
from common import STEP, norms_default

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 120
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L, FR = 2584, 25839
rng = np.random.default_rng(3)


def make_ctx():
    ctx = S.Context(0)
    ctx.db_reset(14)
    for f in range(nf):
        ctx.db_add_synthetic(0x5354525547, f, FR)
    ctx.set_norms(norms_default())
    return ctx


ctxs = [make_ctx() for _ in range(kmax)]
src = ctxs[0].db_read_file(0)
tgts = []
for i in range(16):
    o = int(rng.integers(0, FR - L))
    t = src[o:o + L].copy() + (rng.standard_normal((L, 14)) * 0.05).astype(np.float32)
    t[:, 0] = np.abs(t[:, 0]) + np.float32(0.01)
    tgts.append(np.ascontiguousarray(t))
tdev = [torch.from_numpy(t).to("cuda:0") for t in tgts]
torch.cuda.synchronize()
p = S.SearchParams(10, 1, 0, STEP)
ref = []
for ctx in ctxs:
    for k in range(16):
        ctx.corr_set_target_device(tdev[k].data_ptr(), L, 0.5, 8.0)
        m = S.search(ctx, p)
        if ctx is ctxs[0]:
            ref.append([(x.file, x.start, x.sim) for x in m])
        else:
            assert [(x.file, x.start, x.sim) for x in m] == ref[k]


def worker(ctx, count, start, bar, bad):
    bar.wait()
    for i in range(count):
        k = (start + i) % 16
        ctx.corr_set_target_device(tdev[k].data_ptr(), L, 0.5, 8.0)
        m = S.search(ctx, p)
        if i % 64 == 0 and [(x.file, x.start, x.sim) for x in m] != ref[k]:
            bad.append((start, i))
    ctx.synchronize()


for K in range(1, kmax + 1):
    for rep in range(2):
        bar = threading.Barrier(K + 1)
        bad = []
        ths = [threading.Thread(target=worker, args=(ctxs[j], n, 5 * j, bar, bad)) for j in range(K)]
        for t in ths:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        print(f"{K} in flight: {1e3 * dt / (n * K):.4f} ms per query ({n * K} queries, {len(bad)} wrong)", flush=True)

# the same through the library's own batch call (its threads, no interpreter between the queries)
ptrs = [tdev[(i * 5) % 16].data_ptr() for i in range(n)]
for K in range(1, kmax + 1):
    for rep in range(2):
        t0 = time.perf_counter()
        res = S.search_batch(ctxs[:K], p, ptrs, 0.5, 8.0, target_frames=L)
        dt = time.perf_counter() - t0
        wrong = sum(1 for i in range(0, n, 16) if [(x.file, x.start, x.sim) for x in res[i]] != ref[(i * 5) % 16])
        print(f"search_batch, {K} contexts: {1e3 * dt / n:.4f} ms per query ({n} queries, {wrong} wrong)", flush=True)
# short batches (the threads start with every call)
for nb in (8, 20, 64):
    t0 = time.perf_counter()
    for rep in range(50):
        S.search_batch(ctxs, p, ptrs[:nb], 0.5, 8.0, target_frames=L)
    dt = time.perf_counter() - t0
    print(f"search_batch, {kmax} contexts, batches of {nb}: {1e3 * dt / (50 * nb):.4f} ms per query", flush=True)
