import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import STEP, O, corr_config, make_plants, norms_default, synth_db, synth_target
import strugatzki_amd as S
ctx = S.Context(0)
for num_ch in (14, 31, 32, 33, 48, 63, 64):
  for w in (1.0, 0.5, 0.0):
    for n in (4096, 8192):
        L, frames, ts = 17, [690, 589, 458, 504, 460], 5
        nf = len(frames)
        os.environ["SGZ_FFT_N"] = str(n)
        ctx.set_screen_kernel("spectral")
        db = synth_db(nf, frames, [], seed=1234, num_ch=num_ch)
        tgt = synth_target(ts, L, num_ch=num_ch)
        norms = norms_default(num_ch)
        cfg = corr_config(ts, L, weight=w, normalize=True, max_boost=8.0, num_matches=3, num_ch=num_ch)
        want, sims, boosts = O.correlation_run(cfg, tgt, norms, db, dense=True)
        ctx.db_reset(num_ch)
        for f in db: ctx.db_add_file(f)
        ctx.set_norms(norms)
        ctx.corr_set_target(tgt[ts:ts + L], w, 8.0)
        ctx.corr_run_screen_f16(0, nf)
        worst, hi = 0.0, 0.0
        for f in range(nf):
            gs, _ = ctx.corr_read_dense(f)
            open_ = np.isfinite(sims[f]) & (boosts[f] <= 8.0 * (1 - 1e-4))
            d = gs.astype(np.float64)[open_] - sims[f][open_]
            worst = min(worst, d.min()); hi = max(hi, d.max())
        print(f"ch={num_ch} w={w} n={n}: upper - oracle in [{worst:.3e}, {hi:.3e}] {ctx.screen_kernel_in_use}", flush=True)
