"""What a context's first search costs beyond a cold layout: first search of a fresh context, then the same database
loaded again into the same context (layout from scratch, buffers and code objects already there), then a second
fresh context in the same process (code objects loaded, buffers not)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import strugatzki_amd as S
import bench as B

L, FR, NF = 2584, 25839, 120
gen = S.Context(0)
gen.db_reset(B.NUM_CH)
gen.db_add_synthetic(B.SEED, B.TARGET_FILE, B.TARGET_START + L + 16, [])
tgt = np.ascontiguousarray(gen.db_read_file(0)[B.TARGET_START:B.TARGET_START + L])
gen.close()
params = S.SearchParams(10, 1, 0, B.STEP, capacity=1 << 16)


def load_and_search(ctx, label):
    t0 = time.perf_counter()
    ctx.db_reset(B.NUM_CH)
    for g in range(NF):
        ctx.db_add_synthetic(B.SEED, g, FR, [])
    ctx.set_norms(B.norms_default())
    ctx.synchronize()
    t1 = time.perf_counter()
    ctx.corr_set_target(tgt, 0.5, 8.0)
    t2 = time.perf_counter()
    S.search(ctx, params)
    t3 = time.perf_counter()
    S.search(ctx, params)
    t4 = time.perf_counter()
    print(f"{label}: load {1e3*(t1-t0):.2f}  set_target {1e3*(t2-t1):.2f}  first search {1e3*(t3-t2):.2f}  second {1e3*(t4-t3):.3f} ms", flush=True)


a = S.Context(0)
load_and_search(a, "fresh context, fresh process")
load_and_search(a, "same context, database loaded again")
b = S.Context(0)
load_and_search(b, "second fresh context")
