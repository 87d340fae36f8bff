"""Kernel time of the half-precision screening run on the bench workload (library chosen by SGZ_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import strugatzki_amd as S
from strugatzki_amd import _native as N
import bench as B

L, FR, NF = 2584, 25839, 120
ctx = S.Context(0)
ctx.db_reset(14)
for g in range(NF):
    ctx.db_add_synthetic(B.SEED, g, FR, [])
ctx.set_norms(B.norms_default())
c2 = S.Context(0); c2.db_reset(14); c2.db_add_synthetic(B.SEED, B.TARGET_FILE, B.TARGET_START + L + 16, [])
tgt = c2.db_read_file(0)[B.TARGET_START:B.TARGET_START + L].copy(); c2.close()
ctx.corr_set_target(tgt, 0.5, 8.0)
ms = []
for i in range(12):
    ctx.corr_run_screen_f16(0, NF)
    ctx.synchronize()
    ms.append(ctx.corr_last_kernel_ms())
print(os.environ.get("SGZ_LIB_PATH", "default"), "kernel ms: min %.4f median %.4f" % (min(ms[2:]), sorted(ms[2:])[5]), flush=True)
