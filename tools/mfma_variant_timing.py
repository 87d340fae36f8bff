"""Kernel time of the half-precision screening run for every tile width of the matrix-core kernel
(SGZ_HALF_NB; "model" = the library's own choice), on 120 files / 30 s target and 1500 files / 60 s target.
The bounds a run stores must not depend on the width: their bits are compared with the first variant's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import strugatzki_amd as S
import bench as B

CASES = [(2584, 25839, 120, 300), (5168, 25839, 1500, 60)]
if len(sys.argv) > 1:
    CASES = [CASES[int(sys.argv[1])]]
for L, FR, NF, reps in CASES:
    c2 = S.Context(0); c2.db_reset(14); c2.db_add_synthetic(B.SEED, B.TARGET_FILE, B.TARGET_START + L + 16, [])
    tgt = c2.db_read_file(0)[B.TARGET_START:B.TARGET_START + L].copy(); c2.close()
    first = None
    for nb in ("model", "4", "5", "7", "8", "9", "10"):
        os.environ.pop("SGZ_HALF_NB", None)
        if nb != "model":
            os.environ["SGZ_HALF_NB"] = nb
        ctx = S.Context(0)
        ctx.db_reset(14)
        for g in range(NF):
            ctx.db_add_synthetic(B.SEED, g, FR, [])
        ctx.set_norms(B.norms_default())
        ctx.corr_set_target(tgt, 0.5, 8.0)
        ms = []
        for i in range(reps):
            ctx.corr_run_screen_f16(0, NF)
            ctx.synchronize()
            ms.append(ctx.corr_last_kernel_ms())
        sim, _ = ctx.corr_read_dense(3)
        if first is None:
            first = sim
        tail = sorted(ms[reps // 2:])
        print(f"L={L} files={NF} NB={nb}: kernel ms median {tail[len(tail) // 2]:.4f} min {tail[0]:.4f}"
              f" bounds equal to first: {np.array_equal(first, sim, equal_nan=True)}", flush=True)
        ctx.close()
