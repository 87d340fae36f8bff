"""Compares the half-precision matrix-core screening kernel with the exact fp64 kernel on the bench workload:
error statistics, kernel times and the number of candidates a margin lets through."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import strugatzki_amd as S
from strugatzki_amd import _native as N
import bench as B

L, FR = 2584, 25839
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 40
plants = B.plant_list(10, L, [int(round(j * (NF - 1) / 9)) for j in range(10)], FR)
npl = [N.Plant(p["file"], p["offset"], p["src_file"], p["src_offset"], p["length"], p["amp"], p["gain"]) for p in plants]
ctx = S.Context(0)
ctx.db_reset(14)
for g in range(NF):
    ctx.db_add_synthetic(B.SEED, g, FR, npl)
ctx.set_norms(B.norms_default())
c2 = S.Context(0); c2.db_reset(14); c2.db_add_synthetic(B.SEED, B.TARGET_FILE, B.TARGET_START + L + 16, [])
tgt = c2.db_read_file(0)[B.TARGET_START:B.TARGET_START + L].copy(); c2.close()
ctx.corr_set_target(tgt, 0.5, 8.0)
print("f16 available:", ctx.screen_f16_available, flush=True)
ctx.corr_run(0, NF); ctx.synchronize(); te = ctx.corr_last_kernel_ms()
ex = ctx.corr_read_dense_range(0, NF)
w, fl = ctx.corr_last_work()
for it in range(3):
    ctx.corr_run_screen_f16(0, NF); ctx.synchronize(); th = ctx.corr_last_kernel_ms()
    print(f"exact {te:.3f} ms ({fl/te/1e9:.1f} TF)   f16 {th:.3f} ms ({fl/th/1e9:.1f} TF eq.)", flush=True)
sc = ctx.corr_read_dense_range(0, NF)
es = np.concatenate([a[0] for a in ex]); ss = np.concatenate([a[0] for a in sc])
eb = np.concatenate([a[1] for a in ex]); sb = np.concatenate([a[1] for a in sc])
d = ss.astype(np.float64) - es
print(f"windows {d.size}  max|dsim| {np.max(np.abs(d)):.3e}  rms {np.sqrt(np.mean(d*d)):.3e}  mean {np.mean(d):.3e}  "
      f"max|dboost| {np.max(np.abs(sb-eb)):.2e}")
fin = np.isfinite(d)
print(f"  upper bound: min(U - exact) {np.min(d[fin]):.3e}  median {np.median(d[fin]):.3e}  max {np.max(d[fin]):.3e}  non-finite {np.sum(~fin)}")
hi = es > 0.3
if hi.any():
    print(f"  sim>0.3: n={hi.sum()} max|dsim| {np.max(np.abs(d[hi])):.3e}")
# candidates: threshold = 10th largest per-file maximum of the first 10 files (what the head replay yields)
t0 = np.sort([a[0].max() for a in ex[:10]])[0]
for mg in (0.0, 1e-5, 1e-4, 5e-4, 1e-3, 2e-3, 5e-3, 1e-2):
    print(f"  T0={t0:.4f} margin {mg:g}: exact>{t0-mg:.4f}: {(es > t0 - mg).sum()}   f16: {(ss > t0 - mg).sum()}")
