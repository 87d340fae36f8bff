"""Bounds of the matrix-core kernel for every tile width against the exact similarities (fp64 kernel): every
upper bound must be >= and every lower bound <= the exact value; prints how far the variants are from each other."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import strugatzki_amd as S
import bench as B

L, NF = 2584, 24
rng = np.random.default_rng(5)
frames = [int(x) for x in rng.integers(L, 9 * L, NF)]
frames[3] = L; frames[7] = L + 255; frames[8] = L + 256; frames[9] = L + 2047; frames[10] = L + 2048
c2 = S.Context(0); c2.db_reset(14); c2.db_add_synthetic(B.SEED, B.TARGET_FILE, B.TARGET_START + L + 16, [])
tgt = c2.db_read_file(0)[B.TARGET_START:B.TARGET_START + L].copy(); c2.close()
ref = None
bad = 0
for nb in ("8", "4", "5", "7", "9", "10"):
    os.environ["SGZ_HALF_NB"] = nb
    ctx = S.Context(0)
    ctx.db_reset(14)
    for g in range(NF):
        ctx.db_add_synthetic(B.SEED, g, frames[g], [(g, 100 + g, B.TARGET_FILE, B.TARGET_START, L, 0.05, 1.0)] if g % 5 == 0 and frames[g] >= L + 100 + g else [])
    ctx.set_norms(B.norms_default())
    ctx.corr_set_target(tgt, 0.5, 8.0)
    if ref is None:
        ctx.corr_run(0, NF)
        ref = [ctx.corr_read_dense(f)[0] for f in range(NF)]
    ctx.corr_run_screen_f16(0, NF)
    up = [ctx.corr_read_dense(f)[0] for f in range(NF)]
    lo = [ctx.corr_read_lower(f) for f in range(NF)] if hasattr(ctx, "corr_read_lower") else None
    worst_up = min(float(np.min(u - r)) for u, r in zip(up, ref))
    slack = max(float(np.max(u - r)) for u, r in zip(up, ref))
    if nb == "8":
        base = up
    dv = max(float(np.max(np.abs(u - b))) for u, b in zip(up, base))
    print(f"NB={nb}: min(upper - exact) = {worst_up:.3e}  max(upper - exact) = {slack:.3e}  max |upper - upper(NB=8)| = {dv:.3e}", flush=True)
    bad += worst_up < 0
    ctx.close()
sys.exit(1 if bad else 0)
