"""Compares sgz_corr_run (fp64) with sgz_corr_run_screen (fp32 products) on the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import strugatzki_amd as S
from strugatzki_amd import _native as N
import bench as B

L, FR, NF = 2584, 25839, 40
plants = B.plant_list(10, L, [int(round(j * (NF - 1) / 9)) for j in range(10)], FR)
npl = [N.Plant(p["file"], p["offset"], p["src_file"], p["src_offset"], p["length"], p["amp"], p["gain"]) for p in plants]
ctx = S.Context(0)
ctx.db_reset(14)
for g in range(NF):
    ctx.db_add_synthetic(B.SEED, g, FR, npl)
ctx.set_norms(B.norms_default())
c2 = S.Context(0); c2.db_reset(14); c2.db_add_synthetic(B.SEED, -1, 1000 + L + 16, [])
tgt = c2.db_read_file(0)[1000:1000 + L].copy(); c2.close()
ctx.corr_set_target(tgt, 0.5, 8.0)
for it in range(3):
    ctx.corr_run(0, NF); ctx.synchronize(); te = ctx.corr_last_kernel_ms()
    ex = ctx.corr_read_dense_range(0, NF)
    ctx.corr_run_screen(0, NF); ctx.synchronize(); ts = ctx.corr_last_kernel_ms()
    sc = ctx.corr_read_dense_range(0, NF)
    w, fl = ctx.corr_last_work()
    d = max(float(np.max(np.abs(a[0] - b[0]))) for a, b in zip(ex, sc))
    db = max(float(np.max(np.abs(a[1] - b[1]))) for a, b in zip(ex, sc))
    hi = max(float(np.max(np.abs(a[0] - b[0])[a[0] > 0.5])) if np.any(a[0] > 0.5) else 0.0 for a, b in zip(ex, sc))
    print(f"exact {te:.3f} ms ({fl/te/1e9:.1f} TF)  screen {ts:.3f} ms ({fl/ts/1e9:.1f} TF)  max|dsim| {d:.2e} (sim>0.5: {hi:.2e})  max|dboost| {db:.1e}")
