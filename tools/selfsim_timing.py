"""Self-similarity of a 30-minute file (155 039 feature frames, 13 MFCC + loudness, corrLen 1 s): wall time of the
call; run under rocprofv3 --kernel-trace --stats for the kernel times.  usage: selfsim_timing.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import NUM_CH, SEED, STEP, O, norms_default
import strugatzki_amd as S

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 155039
x = O.synth_file(SEED, 21, frames, NUM_CH)
cfg = O.SelfSimConfig(0, 0, 0, 0, 44100, 1, 0.5, 0, 1.0, 1.0, 0, 1, STEP, NUM_CH - 1)
ext, decim, af_start, _ = O.selfsim_geometry(cfg, frames, frames)
half = O.full_to_feat(44100, STEP)
with S.Context(0) as ctx:
    for _ in range(2):
        t0 = time.perf_counter()
        _, img = ctx.selfsim(x, None, norms_default(), af_start, half, decim, ext, 0.5, want_sim=False)
        dt = time.perf_counter() - t0
        print(f"ext={ext} decim={decim} H={half}: {dt:.3f} s, {ext * (ext + 1) / 2 / dt:.3e} pixel pairs/s", flush=True)
