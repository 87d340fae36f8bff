"""Wall time of extraction calls (upload-bound) for sample counts given on the command line."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import strugatzki_amd as S
rng = np.random.default_rng(0)
ctx = S.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [158_760_000]:
    x16 = rng.integers(-3000, 3000, size=(n, 1), dtype=np.int16)
    xf = x16.astype(np.float32) / np.float32(32768)
    for name, a in (("int16", x16), ("float", xf)):
        ts = []
        for it in range(5):
            t0 = time.perf_counter(); ctx.extract(a, 44100.0); ts.append(time.perf_counter() - t0)
        print(f"n {n} {name} {a.nbytes/1e6:.0f} MB: best {min(ts)*1e3:.2f} ms  median {sorted(ts)[2]*1e3:.2f} ms  -> {a.nbytes/min(ts)/1e9:.1f} GB/s")
