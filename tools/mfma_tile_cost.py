"""Time of a tile of the matrix-core kernel against the number of 32-lag steps (target length): the intercept is
what a tile costs besides its matrix instructions (start, reductions, epilogue).  120 files of 25 839 frames,
tile width 8 forced; tiles per launch and rounds of 512 printed next to the kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SGZ_HALF_NB"] = "8"
import numpy as np
import strugatzki_amd as S
import bench as B

FR, NF, reps = 25839, 120, 300
rows = []
for L in (323, 646, 1292, 2584, 3876, 5168):
    c2 = S.Context(0); c2.db_reset(14); c2.db_add_synthetic(B.SEED, B.TARGET_FILE, B.TARGET_START + L + 16, [])
    tgt = c2.db_read_file(0)[B.TARGET_START:B.TARGET_START + L].copy(); c2.close()
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
    med = sorted(ms[reps // 2:])[reps // 4]
    steps = (L + 15 + 31) // 32
    tiles = NF * -(-(FR - L + 1) // 2048)
    rounds = tiles / 512.0
    rows.append((steps, med * 1e3 / np.ceil(rounds)))
    print(f"L={L} steps={steps} tiles={tiles} rounds={rounds:.2f} kernel {med*1e3:.1f} us -> {med*1e3/np.ceil(rounds):.1f} us per round", flush=True)
    ctx.close()
x = np.array([r[0] for r in rows], dtype=float); y = np.array([r[1] for r in rows])
b, a = np.polyfit(x, y, 1)
print(f"per round: {a:.1f} us + {b:.3f} us per step  (at 82 steps: fixed part {a/(a+82*b)*100:.0f} %)")
