"""Wall time of extraction from 24- and 32-bit samples (decoded on the device by pcm_decode_kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import strugatzki_amd as S
rng = np.random.default_rng(0)
ctx = S.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 158_760_000
for code, width in ((4, 3), (5, 3), (6, 4), (1, 4), (2, 2)):
    raw = rng.integers(0, 256, size=n * width, dtype=np.uint8)
    if code == 1:
        raw = (0.1 * rng.standard_normal(n)).astype(">f4").view(np.uint8)
    ts = []
    for it in range(4):
        t0 = time.perf_counter(); ctx.extract_pcm(raw, code, n, 1, 44100.0); ts.append(time.perf_counter() - t0)
    print(f"layout {code}: {raw.nbytes/1e6:.0f} MB  best {min(ts)*1e3:.2f} ms")
