"""Kernel time of the screening run alone at the headline size (30 s target, 120 files of 5 min), for the library
named by SGZ_LIB_PATH: prints min / median of the launch's HIP-event time."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import strugatzki_amd as S
from common import norms_default, synth_target

L, NF, FR, ts = int(os.environ.get("PROBE_L", 2584)), int(os.environ.get("PROBE_NF", 120)), 25839, 1000
with S.Context(0) as ctx:
    ctx.set_screen_kernel(os.environ.get("PROBE_KERNEL", "spectral"))
    ctx.db_reset(14)
    for f in range(NF):
        ctx.db_add_synthetic(0x5354525547, f, FR)
    ctx.set_norms(norms_default())
    tgt = synth_target(ts, L)
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    for _ in range(300):
        ctx.corr_run_screen_f16(0, NF)
    ctx.synchronize()
    ms = []
    for _ in range(200):
        ctx.corr_run_screen_f16(0, NF)
        ms.append(ctx.corr_last_kernel_ms())
    print(os.path.basename(os.environ.get("SGZ_LIB_PATH", "default")), ctx.screen_kernel_in_use, ctx.screen_geometry,
          "kernel ms min/med %.4f / %.4f" % (min(ms), float(np.median(ms))), flush=True)
