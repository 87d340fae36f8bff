import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import strugatzki_amd as S
from strugatzki_amd import _native as N
from strugatzki_amd.engine import Selector
import bench as B
L, FR, NF = 2584, 25839, 120
plants = B.plant_list(10, L, [int(round(j * (NF - 1) / 9)) for j in range(10)], FR)
npl = [N.Plant(p["file"], p["offset"], p["src_file"], p["src_offset"], p["length"], p["amp"], p["gain"]) for p in plants]
ctx = S.Context(0); ctx.db_reset(14)
for g in range(NF): ctx.db_add_synthetic(B.SEED, g, FR, npl)
ctx.set_norms(B.norms_default())
c2 = S.Context(0); c2.db_reset(14); c2.db_add_synthetic(B.SEED, B.TARGET_FILE, B.TARGET_START + L + 16, [])
tgt = c2.db_read_file(0)[B.TARGET_START:B.TARGET_START + L].copy(); c2.close()
ctx.corr_set_target(tgt, 0.5, 8.0)
for it in range(5):
    t = [time.perf_counter()]
    ctx.corr_run(0, 10); t.append(time.perf_counter())
    ctx.corr_run_screen_f16(10, NF); t.append(time.perf_counter())
    dense = ctx.corr_read_dense_range(0, 10); t.append(time.perf_counter())
    sel = Selector(10, 2, 0, 512, L)
    for g, (s, b) in enumerate(dense):
        sel.feed_dense(g, s, b)
    t.append(time.perf_counter())
    cands = ctx.corr_compact(10, NF, sel.threshold - 1e-6, 1 << 16, rescore=True); t.append(time.perf_counter())
    sel.close()
    names = ["launch exact head", "launch f16 tail", "read dense head", "replay head", "compact + rescore (+ wait)"]
    print(it, " | ".join(f"{n} {1e3*(b-a):.3f}" for n, a, b in zip(names, t, t[1:])), f"| total {1e3*(t[-1]-t[0]):.3f} ms, {len(cands)} candidates", flush=True)
