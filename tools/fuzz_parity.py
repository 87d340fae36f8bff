"""Random search configurations on the GPU against the oracle: match list (file, span) identical, similarities
within one fp32 ulp.  usage: fuzz_parity.py [--big] [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import STEP, O, corr_config, make_plants, norms_default, synth_db, synth_target
import strugatzki_amd as S

BIG = False  # long targets and files of several kernel tiles (slow on the oracle's side: few cases)


def run(cases: int, seed: int, ctx=None, verbose=True) -> int:
    """Runs `cases` random configurations; returns the number of mismatches."""
    rng = np.random.default_rng(seed)
    own = ctx is None
    if own:
        ctx = S.Context(0)
    bad = 0
    for case in range(cases):
        bad += 0 if _one_case(ctx, rng, case, verbose) else 1
        if verbose and case % 20 == 19:
            print(f"{case + 1} cases, {bad} mismatches so far", flush=True)
    if own:
        ctx.close()
    return bad


def _one_case(ctx, rng, case, verbose) -> bool:
    num_ch = int(rng.choice([2, 5, 14, 14, 14, 21, 64]))
    L = int(rng.choice([431, 700, 1000, 2584] if BIG else [1, 2, 3, 16, 17, 40, 96, 150, 431, 700]))
    nf = int(rng.integers(1, 9))
    frames = [int(rng.integers(max(1, L // 2), (12000 if BIG else 4 * L + 900))) for _ in range(nf)]
    ts = int(rng.integers(0, 30))
    num_matches = int(rng.integers(1, 8))
    num_per_file = int(rng.choice([1, 1, 1, 2, 3]))
    min_spacing = int(rng.choice([0, 0, 5 * STEP, 60 * STEP]))
    weight = float(rng.choice([0.0, 0.3, 0.5, 1.0]))
    max_boost = float(rng.choice([1.5, 8.0, 20.0]))
    normalize = bool(rng.random() < 0.85)
    mode = str(rng.choice(["auto", "auto", "f32", "none"]))
    fmax = max(frames)
    plants = make_plants(int(rng.integers(0, 4)), L, nf, fmax, ts)
    plants = [p for p in plants if p["file"] < nf and p["offset"] + L <= frames[p["file"]]]
    db = synth_db(nf, frames, plants, seed=int(rng.integers(1, 1 << 30)), num_ch=num_ch)
    tgt = synth_target(ts, L, num_ch=num_ch)
    for x in db:  # awkward data now and then: flat stretches, large values, offsets, level steps
        kind = rng.integers(0, 8)
        n = x.shape[0]
        if kind == 0 and n > 40:
            a = int(rng.integers(0, n - 20))
            x[a:a + int(rng.integers(10, 400))] = x[a] + (rng.random(x.shape[1]) * 1e-5).astype(np.float32)
        elif kind == 1:
            x *= np.float32(rng.choice([0.01, 30.0, 300.0]))
            x[:, 0] = np.abs(x[:, 0]) + np.float32(0.01)
        elif kind == 2:
            x[:, 1:] += np.linspace(-3.0, 5.0, x.shape[1] - 1, dtype=np.float32)
        elif kind == 3:
            x[n // 2:] += np.float32(2.5)
            x[:, 0] = np.abs(x[:, 0]) + np.float32(0.01)
    norms = norms_default(num_ch) if normalize else None
    cfg = corr_config(ts, L, weight=weight, normalize=normalize, max_boost=max_boost, num_matches=num_matches,
                      num_per_file=num_per_file, min_spacing=min_spacing, num_ch=num_ch)
    want, _, _ = O.correlation_run(cfg, tgt, norms, db, dense=False)
    ctx.db_reset(num_ch)
    for f in db:
        ctx.db_add_file(f)
    ctx.set_norms(norms)
    ctx.corr_set_target(tgt[ts:ts + L], weight, max_boost)
    p = S.SearchParams(num_matches, num_per_file, min_spacing, STEP, screen=mode != "none",
                       screen_mode="auto" if mode == "none" else mode)
    got = S.search(ctx, p)
    ok = [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
    ok = ok and all(abs(g.sim - w.sim) <= 1.2e-7 * max(1.0, abs(w.sim)) or (np.isnan(g.sim) and np.isnan(w.sim))
                    for g, w in zip(got, want))
    if not ok and verbose:
        print(f"MISMATCH case {case}: ch={num_ch} L={L} frames={frames} ts={ts} nm={num_matches} npf={num_per_file} "
              f"sp={min_spacing} w={weight} mb={max_boost} norm={normalize} mode={mode}", flush=True)
        print("  got ", [(g.file, g.start, round(g.sim, 7)) for g in got], flush=True)
        print("  want", [(w.file, w.start, round(w.sim, 7)) for w in want], flush=True)
    return ok


if __name__ == "__main__":
    if "--big" in sys.argv:
        sys.argv.remove("--big")
        BIG = True
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n} cases, {bad} mismatches", flush=True)
    sys.exit(1 if bad else 0)
