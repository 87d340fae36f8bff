"""Step time of the bench workload through the single native call against the same steps issued one by one
from Python, alternating in one process (box-to-box clock differences cancel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import strugatzki_amd as S
from strugatzki_amd import _native as N
from strugatzki_amd.engine import Selector
import bench as B

L, FR, NF = 2584, 25839, 120
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


def native():
    return ctx.corr_search(10, 1, 0, 512)


def stepwise():
    ctx.corr_run_screen_f16(0, NF)
    c = ctx.corr_compact_bounded(0, NF, NF, 1 << 16)
    sel = Selector(10, 1, 0, 512, L)
    assert sel.feed_screened(c, NF, NF)
    out = sel.result()
    sel.close()
    return out


a, b = native(), stepwise()
assert [(m.file, m.start, m.sim) for m in a] == [(m.file, m.start, m.sim) for m in b]
for rnd in range(4):
    for name, fn in (("native", native), ("stepwise", stepwise)):
        for _ in range(5):
            fn()
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(200):
            fn()
        ctx.synchronize()
        print(f"{name}: {(time.perf_counter() - t) / 200 * 1e3:.4f} ms/step", flush=True)
