"""Random search configurations on the GPU against the oracle: match list (file, span) identical, similarities
within one fp32 ulp.  usage: fuzz_parity.py [--big] [--kernels] [cases] [seed]
--kernels: every case also draws the screening kernel (library's choice, spectral with a random transform length,
matrix cores) and, where the oracle's dense run is cheap, checks that the kernel's upper bounds enclose it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import STEP, O, corr_config, make_plants, norms_default, synth_db, synth_target
import strugatzki_amd as S

BIG = False  # long targets and files of several kernel tiles (slow on the oracle's side: few cases)
KERNELS = False  # draw the screening kernel per case


def run(cases: int, seed: int, ctx=None, verbose=True, kernels=None) -> int:
    """Runs `cases` random configurations; returns the number of mismatches."""
    rng = np.random.default_rng(seed)
    own = ctx is None
    if own:
        ctx = S.Context(0)
    draw = KERNELS if kernels is None else kernels
    saved = {k: os.environ.pop(k, None) for k in ("SGZ_FFT_N", "SGZ_SCREEN_KERNEL")}
    bad = 0
    try:
        for case in range(cases):
            bad += 0 if _one_case(ctx, rng, case, verbose, draw) else 1
            if verbose and case % 20 == 19:
                print(f"{case + 1} cases, {bad} mismatches so far", flush=True)
    finally:
        os.environ.pop("SGZ_FFT_N", None)
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
        ctx.set_screen_kernel("auto")
        ctx.set_spectra_format("auto")
    if own:
        ctx.close()
    return bad


def _one_case(ctx, rng, case, verbose, draw_kernel=False) -> bool:
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
    kern, fft_n = "auto", 0
    if draw_kernel:
        kern = str(rng.choice(["auto", "spectral", "spectral", "matrix"]))
        fft_n = int(rng.choice([4096, 8192, 16384]))
        os.environ.pop("SGZ_FFT_N", None)
        if kern == "spectral" and fft_n > L:
            os.environ["SGZ_FFT_N"] = str(fft_n)
        ctx.set_screen_kernel(kern)
        ctx.set_spectra_format(str(rng.choice(["auto", "q16", "f32"])))
    windows = sum(max(f - L + 1, 1) for f in frames)
    dense = draw_kernel and windows * L * num_ch < 3e8
    want, sims, boosts = O.correlation_run(cfg, tgt, norms, db, dense=dense)
    ctx.db_reset(num_ch)
    for f in db:
        ctx.db_add_file(f)
    ctx.set_norms(norms)
    ctx.corr_set_target(tgt[ts:ts + L], weight, max_boost)
    if dense and ctx.screen_f16_available:
        # the screening kernel's value is never below the oracle's similarity where the boost gate is open
        ctx.corr_run_screen_f16(0, nf)
        for f in range(nf):
            gs, _ = ctx.corr_read_dense(f)
            open_ = np.isfinite(sims[f]) & (boosts[f] <= max_boost * (1 - 1e-4))
            d = gs.astype(np.float64)[open_] - sims[f][open_]
            if d.size and not np.all(d >= 0.0):
                if verbose:
                    print(f"BOUND BELOW ORACLE case {case}: file {f} by {d.min():.3e} kern={kern} n={fft_n} "
                          f"in use {ctx.screen_kernel_in_use} {ctx.spectra_format_in_use} {ctx.screen_geometry} ch={num_ch} L={L} frames={frames} "
                          f"w={weight} norm={normalize}", flush=True)
                return False
    p = S.SearchParams(num_matches, num_per_file, min_spacing, STEP, screen=mode != "none",
                       screen_mode="auto" if mode == "none" else mode)
    got = S.search(ctx, p)
    ok = [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want]
    ok = ok and all(abs(g.sim - w.sim) <= 1.2e-7 * max(1.0, abs(w.sim)) or (np.isnan(g.sim) and np.isnan(w.sim))
                    for g, w in zip(got, want))
    if not ok and verbose:
        print(f"MISMATCH case {case}: ch={num_ch} L={L} frames={frames} ts={ts} nm={num_matches} npf={num_per_file} "
              f"sp={min_spacing} w={weight} mb={max_boost} norm={normalize} mode={mode} kern={kern} n={fft_n}", flush=True)
        print("  got ", [(g.file, g.start, round(g.sim, 7)) for g in got], flush=True)
        print("  want", [(w.file, w.start, round(w.sim, 7)) for w in want], flush=True)
    return ok


if __name__ == "__main__":
    if "--big" in sys.argv:
        sys.argv.remove("--big")
        BIG = True
    if "--kernels" in sys.argv:
        sys.argv.remove("--kernels")
        KERNELS = True
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n} cases, {bad} mismatches", flush=True)
    sys.exit(1 if bad else 0)
