import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import strugatzki_amd as S
from strugatzki_amd import _native as N
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
for npf, sp in ((1, 0), (2, 0), (2, 512 * 100), (5, 512 * 10)):
    fn = lambda: ctx.corr_search(10, npf, sp, 512)
    for _ in range(20): fn()
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(100): fn()
    ctx.synchronize()
    print(f"numPerFile {npf} minSpacing {sp}: {(time.perf_counter() - t) / 100 * 1e3:.3f} ms/search", flush=True)
