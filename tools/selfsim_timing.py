"""Times sgz_selfsim_run at (a fraction of) BASELINE config 4: 30 min file, corrLen 44100 (H = 86), auto-decimation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import strugatzki_amd as S
from common import O, SEED, NUM_CH, norms_default

frames_n = int(sys.argv[1]) if len(sys.argv) > 1 else 155039
H = 86
x = O.synth_file(SEED, 7, frames_n, NUM_CH)
num_corrs = frames_n - 2 * H + 1
decim = 1 if num_corrs <= 0xB504 else (num_corrs + 0xB503) // 0xB504
ext = num_corrs // decim
ctx = S.Context(0)
t0 = time.perf_counter()
sim, img = ctx.selfsim(x, None, norms_default(), 0, H, decim, ext, 0.5, want_sim=False, want_img=True)
dt = time.perf_counter() - t0
pairs = ext * (ext + 1) // 2
print(f"frames {frames_n} decim {decim} imgExt {ext} pairs {pairs:.3e}  wall {dt:.3f} s  {pairs/dt:.3e} pairs/s")
d = np.diag(img[::-1])
print("diagonal all white:", bool(np.all(d == 0xFFFFFF)), " symmetric:", bool(np.array_equal(img[::-1], img[::-1].T)))
