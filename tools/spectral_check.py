"""Spectral screening against the oracle (small cases) and against the matrix-core kernel (config-3 size)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import strugatzki_amd as S
from common import O, STEP, corr_config, make_plants, norms_default, synth_db, synth_target

def small():
    for L, frames, nf in [(64, 400, 3), (431, 2583, 3), (3300, 5000, 3), (16, 100, 2), (33, 33, 2), (100, 4500, 3), (1200, 9000, 2), (2584, 20000, 1)]:
        ts = 20
        db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
        tgt = synth_target(ts, L)
        norms = norms_default()
        _, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=3), tgt, norms, db)
        for kern in ("matrix", "spectral"):
            with S.Context(0) as ctx:
                ctx.set_screen_kernel(kern)
                ctx.db_reset(db[0].shape[1])
                for f in db: ctx.db_add_file(f)
                ctx.set_norms(norms)
                ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
                ctx.corr_run_screen_f16(0, nf)
                lo, hi = 1e9, -1e9
                for f in range(nf):
                    gs, gb = ctx.corr_read_dense(f)
                    open_ = boosts[f] <= 8.0 * (1 - 1e-4)
                    d = gs.astype(np.float64) - sims[f]
                    if open_.any():
                        lo = min(lo, d[open_].min()); hi = max(hi, d[open_].max())
                    assert np.allclose(gb, boosts[f], rtol=1e-5)
                print(f"L={L} frames={frames} {kern:8s} in use={ctx.screen_kernel_in_use}: upper - oracle in [{lo:.3e}, {hi:.3e}]", flush=True)

def big():
    L, NF, FR, ts = 2584, 120, 25839, 1000
    for kern in ("matrix", "spectral"):
        with S.Context(0) as ctx:
            ctx.set_screen_kernel(kern)
            ctx.db_reset(14)
            for f in range(NF): ctx.db_add_synthetic(0x5354525547, f, FR)
            ctx.set_norms(norms_default())
            tgt = synth_target(ts, L)
            ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
            t0 = time.perf_counter(); ctx.corr_run_screen_f16(0, NF); ctx.synchronize(); cold = time.perf_counter() - t0
            ms = []
            for _ in range(30):
                ctx.corr_run_screen_f16(0, NF); ctx.synchronize(); ms.append(ctx.corr_last_kernel_ms())
            ups = np.concatenate([ctx.corr_read_dense(f)[0] for f in range(4)])
            print(f"{kern}: in use {ctx.screen_kernel_in_use}, cold {cold*1e3:.2f} ms, kernel ms min/med {min(ms):.4f}/{np.median(ms):.4f}", flush=True)
            if kern == "matrix": ref = ups
            else: print("spectral upper - matrix upper: min %.3e max %.3e" % ((ups - ref).min(), (ups - ref).max()))
            t0 = time.perf_counter()
            for _ in range(200): got = S.search(ctx, S.SearchParams(10, 1, 0, STEP))
            dt = (time.perf_counter() - t0) / 200
            print(f"   search {dt*1e3:.4f} ms/step", [(g.file, g.start // STEP, round(g.sim, 5)) for g in got][:4], flush=True)

if __name__ == "__main__":
    small()
    big()
