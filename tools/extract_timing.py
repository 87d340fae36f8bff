"""Times sgz_extract on the extraction configuration of the benchmark set: 1 h mono 44.1 kHz, fftSize 1024, hop 512, 13 MFCC + loudness."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import strugatzki_amd as S
from strugatzki_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 158_760_000
rng = np.random.default_rng(0)
x = (0.1 * rng.standard_normal(n, dtype=np.float32))[:, None]
ctx = S.Context(0)
for it in range(3):
    t0 = time.perf_counter()
    f = ctx.extract(x, 44100.0)
    dt = time.perf_counter() - t0
    k = N.lib.sgz_last_kernel_ms(ctx._h)
    print(f"frames {f.shape[0]}  wall {dt*1e3:.1f} ms (incl. H2D of {x.nbytes/1e6:.0f} MB)  kernel {k:.3f} ms "
          f"= {n/44100.0/(k*1e-3):.3e} x realtime, {(x.nbytes + f.nbytes)/k/1e6:.1f} GB/s algorithmic")
# the same hour as 16-bit samples, decoded on the device (what a 16-bit audio file goes up as)
x16 = np.round(x * 32767).astype(np.int16)
for it in range(3):
    t0 = time.perf_counter()
    f = ctx.extract(x16, 44100.0)
    dt = time.perf_counter() - t0
    k = N.lib.sgz_last_kernel_ms(ctx._h)
    print(f"int16: frames {f.shape[0]}  wall {dt*1e3:.1f} ms (incl. H2D of {x16.nbytes/1e6:.0f} MB)  kernel {k:.3f} ms")
# twenty minutes of stereo, as floats and as 16-bit samples
xs = (0.1 * rng.standard_normal((52_920_000, 2), dtype=np.float32))
for name, a in (("stereo float", xs), ("stereo int16", np.round(xs * 32767).astype(np.int16))):
    for it in range(2):
        f = ctx.extract(a, 44100.0)
        k = N.lib.sgz_last_kernel_ms(ctx._h)
    print(f"{name}: frames {f.shape[0]}  kernel {k:.3f} ms")
