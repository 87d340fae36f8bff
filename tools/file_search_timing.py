"""The search through the file-based boundary (sgz_run_correlation_xml over a database folder of feature AIFFs):
first call (files read, decoded, uploaded) against later calls (database resident on the context), for 120 files
of 5 minutes and a 30 s target.  Files are written to TMPDIR (tmpfs or disk as the box has it)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import strugatzki_amd as S
from strugatzki_amd import processors as P
from strugatzki_amd.processors import FeatureCorrelation, FeatureExtraction, Span
import bench as B

NF, FR, L, STEP = 120, 25839, 2584, 512
RATE = 44100.0 / STEP


def entry(folder, name, frames):
    feat = os.path.join(folder, f"{name}_feat.aif")
    meta = os.path.join(folder, f"{name}_feat.xml")
    P.write_float_aiff(feat, frames, RATE)
    open(meta, "w").write(FeatureExtraction.Config(os.path.join(folder, f"{name}.aif"), feat, meta, 13, 1024, 2, 0).to_xml())
    return meta


with tempfile.TemporaryDirectory() as d:
    gen = S.Context(0)

    def synth(file_id, frames):
        gen.db_reset(14)
        gen.db_add_synthetic(B.SEED, file_id, frames, [])
        return gen.db_read_file(0).copy()

    tgt = synth(B.TARGET_FILE, B.TARGET_START + L + 16)
    for g in range(NF):
        x = synth(g, FR)
        if g == 7:
            x[3000:3000 + L] = tgt[B.TARGET_START:B.TARGET_START + L]
        entry(d, f"snd{g:03d}", x)
    gen.close()
    in_meta = entry(d, "input", tgt)
    P.write_float_aiff(os.path.join(d, "feat_norms.aif"), np.ascontiguousarray(B.norms_default().T), RATE)
    cfg = FeatureCorrelation.Config(d, in_meta, FeatureCorrelation.Punch(Span(B.TARGET_START * STEP, (B.TARGET_START + L) * STEP), 0.5),
                                    None, 22050, 88200, True, 8.0, 10, 1, 0)
    wins = NF * (FR - L + 1)
    with S.Context(0) as ctx:
        for i in range(5):
            t0 = time.perf_counter()
            res = FeatureCorrelation.run(ctx, cfg)
            dt = time.perf_counter() - t0
            print(f"call {i}: {dt * 1e3:.2f} ms  {wins / dt:.3e} windows/s  best {res[0].sim:.4f} in "
                  f"{os.path.basename(res[0].file)} at {res[0].punch.start // STEP}", flush=True)
