"""Random punch-in / punch-out searches on the GPU against the oracle.  usage: fuzz_punch.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import STEP, O, make_plants, norms_default, punch_config, synth_db, synth_target
import strugatzki_amd as S


def run(cases: int, seed: int, ctx=None, verbose=True) -> int:
    rng = np.random.default_rng(seed)
    own = ctx is None
    if own:
        ctx = S.Context(0)
    bad = 0
    for case in range(cases):
        Lin, Lout = int(rng.choice([8, 24, 48, 100])), int(rng.choice([8, 32, 60]))
        gap = int(rng.integers(0, 40))
        ts = int(rng.integers(0, 20))
        nf = int(rng.integers(1, 7))
        span = Lin + gap + Lout
        frames = [int(rng.integers(span + 5, span + 700)) for _ in range(nf)]
        num_matches, num_per_file = int(rng.integers(1, 7)), int(rng.choice([1, 1, 2, 3]))
        min_spacing = int(rng.choice([0, 0, 10 * STEP, 40 * STEP]))
        min_punch = int(rng.integers(0, gap + Lin + 10))
        max_punch = min_punch + int(rng.integers(0, 60))
        w_in, w_out = float(rng.choice([0.0, 0.5, 1.0])), float(rng.choice([0.25, 0.75]))
        plants = [p for p in make_plants(int(rng.integers(0, 4)), span, nf, max(frames), ts)
                  if p["file"] < nf and p["offset"] + span <= frames[p["file"]]]
        db = synth_db(nf, frames, plants, seed=int(rng.integers(1, 1 << 30)))
        for x in db:
            if rng.integers(0, 6) == 0 and x.shape[0] > 60:
                a = int(rng.integers(0, x.shape[0] - 30))
                x[a:a + int(rng.integers(10, 200))] = x[a]
        tgt = synth_target(ts, span)
        norms = norms_default()
        o_start = ts + Lin + gap
        cfg = punch_config(ts, Lin, o_start, Lout, min_punch=min_punch, max_punch=max_punch, out_weight=w_out,
                           weight=w_in, num_matches=num_matches, num_per_file=num_per_file, min_spacing=min_spacing)
        want, _, _ = O.correlation_run_punch(cfg, tgt, norms, db)
        ctx.db_reset(db[0].shape[1])
        for f in db:
            ctx.db_add_file(f)
        ctx.set_norms(norms)
        got = S.search_punch(ctx, S.SearchParams(num_matches, num_per_file, min_spacing, STEP), tgt[ts:ts + Lin], w_in,
                             tgt[o_start:o_start + Lout], w_out, 8.0, min_punch, max_punch)
        ok = [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
        ok = ok and all(abs(g.sim - w.sim) <= 1e-5 or (np.isnan(g.sim) and np.isnan(w.sim)) for g, w in zip(got, want))
        if not ok:
            bad += 1
            if verbose:
                print(f"MISMATCH case {case}: Lin={Lin} Lout={Lout} gap={gap} ts={ts} frames={frames} nm={num_matches} "
                      f"npf={num_per_file} sp={min_spacing} punch={min_punch}..{max_punch} w={w_in}/{w_out}", flush=True)
                print("  got ", [(g.file, g.start, g.stop, round(g.sim, 6)) for g in got], flush=True)
                print("  want", [(w.file, w.start, w.stop, round(w.sim, 6)) for w in want], flush=True)
    if own:
        ctx.close()
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n} cases, {bad} mismatches", flush=True)
    sys.exit(1 if bad else 0)
