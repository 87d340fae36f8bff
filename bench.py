#!/usr/bin/env python3
"""Benchmark of the FeatureCorrelation hot path on MI355X.

One step = one complete database search on device-resident synthetic feature
files: normalise + window statistics + sliding correlation of every window on
the GPU, then the reference's match selection (head replay, candidate
compaction, tail replay; one all-gather when sharded).  Metric: candidate
windows per second, whole job.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Workload at N=1: the configuration the headline metric is quoted on -- 30 s target (2584 feature frames)
against a 10 h database (120 files x 300 s), 13 MFCC + loudness,
temporalWeight 0.5, top-10.  For N>1 the default is the large-database configuration:
60 s target (5168 frames) against 1500 files x 300 s per GPU (125 h per GPU,
1000 h on 8 GPUs), sharded by file; --target-seconds / --files-per-gpu override
either.  The bounds of every window come from the spectral kernel (fp32 block
transforms, HBM-bound) where the target is long enough, else from the matrix-core
kernel; --screen-kernel forces one.  The same run also times the search with the
other screening kernel (roofline.matrix_cores) and the reference-precision dense
path (fp64 kernel over every window, roofline.exact).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x5354525547
NUM_CH = 14
STEP = 512
TARGET_FILE = -1
TARGET_START = 1000

# peaks (MI355X): fp64 vector == fp64 matrix rate, 256 CU x 128 flop/clk x 2.4 GHz (spec 78.6 TFLOP/s);
# HBM3E 8 TB/s spec
PEAK_F64_TFLOPS = 78.6
# fp32: vector FMA and f32-input MFMA share the same 157.3 TFLOP/s peak
PEAK_F32_TFLOPS = 157.3
PEAK_F16_TFLOPS = 2516.6  # dense f16/bf16 MFMA: 256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0
SUSTAINED_F16_RANDOM_TFLOPS = 1965.0  # measured, tools/mfma_peak.hip (register operands, random data, 20-40 ms runs)


def plant_list(num: int, L: int, files: list[int], frames_per_file: int):
    """Copy j of the target goes into files[j] at a distinct offset, noise amplitude 0.02*(j+1)."""
    out = []
    for j in range(num):
        off = (4001 + 1733 * j) % (frames_per_file - L)
        out.append(dict(file=files[j], offset=off, src_file=TARGET_FILE, src_offset=TARGET_START, length=L,
                        amp=float(np.float32(0.02) * (j + 1)), gain=1.0))
    return out


def norms_default() -> np.ndarray:
    n = np.zeros((NUM_CH, 2), dtype=np.float32)
    n[0] = (0.0, 1.5)
    for c in range(1, NUM_CH):
        n[c] = (np.float32(0.01) * c, np.float32(1.0) + np.float32(0.02) * c)
    return n


def cpu_baseline(L: int, frames_per_file: int, budget_s: float) -> dict:
    """Times the single-threaded CPU oracle on a bounded sample of the same workload."""
    from oracle import oracle_lib as O

    tgt = O.synth_file(SEED, TARGET_FILE, TARGET_START + L + 16, NUM_CH)
    cfg = O.CorrConfig(TARGET_START * STEP, (TARGET_START + L) * STEP, 0.5, 1, 8.0, 10, 1, 0, STEP, NUM_CH - 1)
    norms = norms_default()
    # calibrate on a short file, then size the sample for ~budget_s of CPU work
    probe_frames = L + 400
    db = [O.synth_file(SEED, 0, probe_frames, NUM_CH)]
    t0 = time.perf_counter()
    O.correlation_run(cfg, tgt, norms, db, dense=False)
    dt = time.perf_counter() - t0
    rate = (probe_frames - L + 1) / dt
    want = max(1000, int(rate * budget_s))
    frames = min(frames_per_file, L + want - 1)
    nfiles = max(1, min(8, int(round(want / (frames - L + 1)))))
    db = [O.synth_file(SEED, f, frames, NUM_CH) for f in range(nfiles)]
    t0 = time.perf_counter()
    O.correlation_run(cfg, tgt, norms, db, dense=False)
    dt = time.perf_counter() - t0
    windows = nfiles * (frames - L + 1)
    return {
        "value": windows / dt,
        "unit": "candidate windows/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{nfiles} file(s) x {frames} frames, target {L} frames = {windows} windows in {dt:.1f} s "
                  f"(C restatement of the reference loop, gcc -O2, one thread)",
    }


def pmc_traffic_bytes(kernel: str, summary: str = "r02_pmc_summary.csv"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of the default N=1 command
    (profiles/<summary>): 2 x FETCH_SIZE (gfx950 tallies 128-byte read requests at 64 bytes) +
    WRITE_SIZE, both in KiB.  None when the summary does not hold the kernel."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", summary)
    try:
        import csv
        vals = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["kernel"] == kernel.split("<")[0] and row["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    vals[row["counter"]] = float(row["mean_per_dispatch"])
        if len(vals) == 2:
            return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    except OSError:
        pass
    return None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is less than a tenth of a millisecond: the defaults keep the GPU busy for about two seconds (long
    # enough for the clocks to settle and for a once-a-second utilisation sampler to see the run) and still finish
    # at once
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--files-per-gpu", type=int, default=None,
                    help="database files per GPU (default: 120 at N=1, 1500 at N>1)")
    ap.add_argument("--file-seconds", type=float, default=300.0)
    ap.add_argument("--target-seconds", type=float, default=None,
                    help="target length (default: 30 at N=1, 60 at N>1)")
    ap.add_argument("--exact-steps", type=int, default=100,
                    help="steps of the reference-precision search (fp64 kernel, every window) timed after the main "
                         "loop at N=1 for roofline.exact; 0 disables")
    ap.add_argument("--num-matches", type=int, default=10)
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed steps for about this long (counted as 0.3 ms each) before the warm-up steps, so that the "
                         "clocks have settled")
    ap.add_argument("--timing-every", type=int, default=32,
                    help="HIP events with one launch in n of the dominant kernel (matrix-core path).  A launch that "
                         "carries events costs the host about 20 us more than one that does not (measured: "
                         "tools/search_loop.cpp), a tenth of a step at the default workload, so the kernel's average "
                         "duration over the timed region comes from a sample of its launches")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget; 0 disables")
    ap.add_argument("--exact-only", action="store_true",
                    help="fp64 kernel for every file (no screening + re-scoring)")
    ap.add_argument("--sharded", action="store_true",
                    help="take the multi-GPU code path (RCCL group, candidate exchange) even with one rank")
    ap.add_argument("--screen-mode", choices=["auto", "f16", "f32"], default="auto",
                    help="screening path: certified bounds from the spectral / matrix-core kernels (f16) or the "
                         "fp32-product kernel (f32)")
    ap.add_argument("--screen-kernel", choices=["auto", "matrix", "spectral"], default="auto",
                    help="kernel behind the certified bounds: fp32 block transforms (spectral, HBM-bound), "
                         "half-precision products on the matrix cores (matrix), or the library's choice")
    ap.add_argument("--other-steps", type=int, default=2000,
                    help="steps of the same search with the screening kernel the library did not pick, timed after "
                         "the main loop at N=1 (roofline.matrix_cores / roofline.spectral); 0 disables")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import strugatzki_amd as S
    from strugatzki_amd import _native as N
    from strugatzki_amd.search import shard_files

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.files_per_gpu is None:
        args.files_per_gpu = 120 if world == 1 else 1500
    if args.target_seconds is None:
        args.target_seconds = 30.0 if world == 1 else 60.0
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py --gpus N>1 must be launched with torch.distributed.run", file=sys.stderr)
            return 2
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (no CPU fallback)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.sharded
    # stdout carries the one JSON line: RCCL prints a version banner there when it starts
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if sharded:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    sr = 44100.0
    L = int((int(args.target_seconds * sr + 0.5) + TARGET_START * STEP + STEP // 2) // STEP) - TARGET_START
    frames_per_file = -(-int(args.file_seconds * sr) // STEP) - 1  # ceil(S/step) - 1
    head_n = args.num_matches
    # weak scaling: every rank holds files_per_gpu files of the database, dealt round robin
    total_files = world * args.files_per_gpu
    # plants: spread over the whole database, first one inside the head
    plant_files = [int(round(j * (total_files - 1) / max(args.num_matches - 1, 1))) for j in range(args.num_matches)]
    plants = plant_list(args.num_matches, L, plant_files, frames_per_file)
    nplants = [N.Plant(p["file"], p["offset"], p["src_file"], p["src_offset"], p["length"], p["amp"], p["gain"])
               for p in plants]

    # sharded: the library's kernels and the RCCL exchange share one stream (no host wait between them)
    ctx = S.Context.on_torch_stream(local_rank) if sharded else S.Context(local_rank)
    ctx.db_reset(NUM_CH)
    if sharded:
        # numPerFile = 1: no file needs to be replicated (every rank finds the exact maxima of its own files)
        head_global = []
        tail_global = [g for g in range(total_files) if g % world == rank]
    else:
        head_global = list(range(min(head_n, total_files)))
        tail_global = list(range(len(head_global), total_files))
    for g in head_global + tail_global:
        ctx.db_add_synthetic(SEED, g, frames_per_file, nplants)
    ctx.set_norms(norms_default())
    # the target: frames [TARGET_START, TARGET_START + L) of the dedicated target file
    ctx_t = S.Context(local_rank)
    ctx_t.db_reset(NUM_CH)
    ctx_t.db_add_synthetic(SEED, TARGET_FILE, TARGET_START + L + 16, [])
    target = ctx_t.db_read_file(0)[TARGET_START:TARGET_START + L].copy()
    ctx_t.close()
    ctx.corr_set_target(target, 0.5, 8.0)
    ctx.set_screen_kernel(args.screen_kernel)

    params = S.SearchParams(args.num_matches, 1, 0, STEP, capacity=1 << 16, screen=not args.exact_only,
                            screen_mode=args.screen_mode)
    mode = "f64" if args.exact_only else ("f16" if args.screen_mode != "f32" and ctx.screen_f16_available else "f32")
    unique_windows = total_files * (frames_per_file - L + 1)

    def step():
        if sharded:
            return S.search_sharded(ctx, params, head_global, tail_global, num_files_total=total_files)
        return S.search(ctx, params)

    def sync():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    # Host side hygiene of a long-running search service: the interpreter's full
    # garbage collections walk every object torch created at import (tens of ms);
    # park those objects in the permanent generation before the loop.
    import gc
    gc.collect()
    gc.freeze()

    # HIP events travel with one launch in --timing-every of the dominant kernel
    timing_every = max(1, args.timing_every) if mode == "f16" else 1
    ctx.set_kernel_timing(timing_every)
    # Clock settling, untimed and reported as such: after an idle period the chip needs some tens of
    # milliseconds of load before it holds its clocks; the first steps then run 5-10 % slower than the rest.
    # (a fixed count, the same on every rank: the sharded step holds a collective)
    matches = step()
    # which kernel the layout of this database and target uses for the bounds
    spectral = mode == "f16" and ctx.screen_kernel_in_use == "spectral"
    fft_n, fft_hop, fft_blocks = ctx.screen_geometry
    spectra_q16 = ctx.spectra_format_in_use == "q16"
    settle_steps = int(args.settle_ms / 0.3)
    for _ in range(settle_steps):
        matches = step()
    for _ in range(args.warmup):
        matches = step()
    kernel_ms = []
    sync()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        matches = step()
        step_ms.append((time.perf_counter() - ts) * 1e3)
        if mode != "f16":  # several runs per step: the tail launch is the last one
            kernel_ms.append(ctx.corr_last_kernel_ms())
    if os.environ.get("SGZ_DEBUG"):
        print("step ms:", " ".join(f"{x:.2f}" for x in step_ms), file=sys.stderr)
    sync()
    elapsed = time.perf_counter() - t0
    # HIP events travel with every launch of the dominant kernel in the timed region; they are read here, after
    # it (a host-side wait on them between steps would sit inside the timed region)
    if mode == "f16":  # one run per step
        kernel_ms = [float(x) for x in ctx.kernel_ms_history(args.steps // timing_every) if x > 0]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    screened_work = ctx.corr_last_work()

    # Cold step: the half-precision image, prefix records and centring constants of the database are built by the
    # first search after the database, the norms or the target length change and kept afterwards (the timed steps
    # above are warm: a service answering queries against a resident database).  Here other norms and then the
    # right ones again are set before each step, which discards that preparation (and the layout tables; setting
    # the norms already in force changes nothing).
    cold_ms = []
    if not sharded:
        other = norms_default().copy()
        other[:, 1] += 1.0
        for _ in range(5):
            ctx.set_norms(other)
            ctx.set_norms(norms_default())
            ctx.corr_set_target(target, 0.5, 8.0)
            ctx.synchronize()
            tc = time.perf_counter()
            step()
            cold_ms.append((time.perf_counter() - tc) * 1e3)

    # The same search with the other screening kernel, timed in the same run (the layout is rebuilt, so the first
    # steps are cold and untimed).
    alt = None
    if world == 1 and not sharded and mode == "f16" and args.other_steps > 0 and args.screen_kernel == "auto":
        other_name = "matrix" if spectral else "spectral"
        ctx.set_screen_kernel(other_name)
        ctx.set_kernel_timing(8)
        for _ in range(100):
            om = step()
        if ctx.screen_kernel_in_use == other_name:
            ctx.synchronize()
            to = time.perf_counter()
            for _ in range(args.other_steps):
                om = step()
            ctx.synchronize()
            o_elapsed = time.perf_counter() - to
            o_ms = [float(x) for x in ctx.kernel_ms_history(args.other_steps // 8) if x > 0]
            o_windows, o_flops = ctx.corr_last_work()
            alt = (other_name, o_elapsed / args.other_steps, float(np.mean(o_ms)) if o_ms else 0.0, o_windows, o_flops,
                     [(m.file, m.start, m.stop, m.sim) for m in om] == [(m.file, m.start, m.stop, m.sim) for m in matches],
                     ctx.screen_geometry)
        ctx.set_screen_kernel(args.screen_kernel)
        ctx.set_kernel_timing(timing_every)
        step()

    # The same search at reference arithmetic: the fp64 kernel evaluates every window (no screening), the
    # selection replays over its values.  Timed here, after the main loop, so that the driver's own run carries
    # the number next to the screened one.
    exact = None
    if world == 1 and not sharded and not args.exact_only and args.exact_steps > 0:
        pe = S.SearchParams(args.num_matches, 1, 0, STEP, capacity=1 << 16, screen=False)
        ctx.set_kernel_timing(1)
        for _ in range(3):
            em = S.search(ctx, pe)
        ctx.synchronize()
        e_ms = []
        te = time.perf_counter()
        for _ in range(args.exact_steps):
            em = S.search(ctx, pe)
            e_ms.append(ctx.corr_last_kernel_ms())  # the tail launch: all files but the first num_matches
        ctx.synchronize()
        e_elapsed = time.perf_counter() - te
        e_windows, e_flops = ctx.corr_last_work()
        ek = float(np.mean(e_ms))
        exact = {
            "kernel": "corr_kernel<double>",
            "dtype": "f64",
            "windows_per_s": unique_windows / (e_elapsed / args.exact_steps),
            "ms_per_step": e_elapsed / args.exact_steps * 1e3,
            "steps": args.exact_steps,
            "kernel_ms": ek,
            "windows_per_launch": e_windows,
            "achieved": e_flops / (ek * 1e-3) / 1e12 if ek > 0 else 0.0,
            "peak": PEAK_F64_TFLOPS,
            "unit": "TFLOP/s",
            "frac": (e_flops / (ek * 1e-3) / 1e12) / PEAK_F64_TFLOPS if ek > 0 else 0.0,
            "verified": [(m.file, m.start // STEP) for m in em] == [(p["file"], p["offset"]) for p in plants],
            "same_matches_as_screened": [(m.file, m.start, m.stop) for m in em] == [(m.file, m.start, m.stop) for m in matches],
            "note": "every window evaluated by the fp64 kernel (v_fmac_f64, fp64 vector peak 78.6 TFLOP/s), selection "
                    "replayed over its values: the search at the reference's arithmetic, no screening",
        }

    # known answer: match j is plant j (noise grows with j), at its planted offset
    got = [(m.file, m.start // STEP) for m in matches]
    want = [(p["file"], p["offset"]) for p in plants]
    verified = got == want

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # dominant kernel: the launch that evaluates every window of the step
        tail_windows, tail_flops = screened_work
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = tail_windows * 64.0  # one new frame in (56 B), similarity and boost out (8 B)

        def matrix_roofline(kms, flops):
            ach = flops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
            return {
                "bound": "mfma", "kernel": "corr_mfma_kernel", "achieved": ach, "peak": PEAK_F16_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / PEAK_F16_TFLOPS,
                # what tools/mfma_peak.hip sustains with this instruction shape, register operands only, on
                # pseudo-random half-precision data (the chip's clock under matrix load depends on operand bit
                # activity: 2480 TFLOP/s on zeros, 2320 on smooth small numbers, 1965 on random data)
                "sustained_peak_random_operands": SUSTAINED_F16_RANDOM_TFLOPS,
                "frac_of_sustained": ach / SUSTAINED_F16_RANDOM_TFLOPS,
                "note": "one launch over all files: v_mfma_f32_16x16x32_f16 on half-precision copies of target and "
                        "database (dense f16 MFMA peak 2516.6 TFLOP/s = 256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz), fp32 "
                        "sums, fp64 window statistics; it yields certified bounds, and every window the bounds cannot "
                        "exclude is re-evaluated in fp64 from the raw frames before the selection replay",
            }

        def spectral_roofline(kms, windows, geom):
            n, hop, blocks = geom
            ach = windows * 64.0 / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
            # what the kernel is built to move: the blocks' spectra and 16 B of cached statistics per window in (20 B in
            # blocks where some window's boost comes near maxBoost: none in this workload); out only the list of the
            # few windows that can hold their file's maximum and one bound per file
            q16 = spectra_q16
            # (16-bit spectra: 4 bytes per complex bin and a 4-byte scale per band of 64 bins; every workgroup also
            # reads the target's spectra, which the caches serve)
            moved = blocks * (2.0 * NUM_CH * n + NUM_CH * n / 32.0 if q16 else 4.0 * NUM_CH * n) + windows * 16.0
            return {
                "bound": "hbm", "kernel": "corr_fft_kernel", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ach / PEAK_HBM_GBS,
                "algorithmic_bytes": windows * 64.0,
                "bytes_by_design": moved,
                "gbs_by_design": moved / (kms * 1e-3) / 1e9 if kms > 0 else 0.0,
                "transform_length": n, "windows_per_block": hop, "blocks": blocks,
                "spectra": "16-bit integers, one scale per 64 bins" if q16 else "fp32",
                "note": "one launch over all files, one workgroup per block of N frames: the cross terms of a block's "
                        "N - L + 1 windows are one circular correlation, evaluated from spectra of the blocks kept "
                        "in HBM with the database (16-bit integers with a scale per band, whose measured distance to "
                        "the fp32 spectra is part of the certified error; product with the target's spectra, channels "
                        "summed in the frequency domain, one fp32 inverse transform in LDS); achieved = 64 algorithmic bytes per window (one new "
                        "frame in, similarity and boost out) over the kernel's duration, against 8 TB/s; "
                        "bytes_by_design is what this formulation has to move (overlapping blocks, cached window "
                        "statistics; the search reads candidate lists, not an upper bound per window).  It yields certified bounds, and every "
                        "window the bounds cannot exclude is re-evaluated in fp64 from the raw frames before the "
                        "selection replay",
            }

        is_c3 = (L, total_files) == (2584, 120)
        if spectral:
            roof = spectral_roofline(k_ms, tail_windows, (fft_n, fft_hop, fft_blocks))
            roof["traffic"] = pmc_traffic_bytes("corr_fft_kernel") if is_c3 else None
        elif mode == "f16":
            roof = matrix_roofline(k_ms, tail_flops)
            roof["traffic"] = pmc_traffic_bytes("corr_mfma_kernel") if is_c3 else None
        else:
            achieved = tail_flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
            peak = {"f64": PEAK_F64_TFLOPS, "f32": PEAK_F32_TFLOPS}[mode]
            roof = {
                "bound": "mfma", "kernel": {"f64": "corr_kernel<double>", "f32": "corr_kernel<float>"}[mode],
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                "note": {
                    "f64": "fp64 FMA on the vector pipe; MI355X fp64 vector and matrix peaks are both 78.6 TFLOP/s",
                    "f32": "tail launch: fp32 products / partial sums on the vector pipe (157.3 TFLOP/s peak, the same as "
                           "the f32-input MFMA peak), fp64 window statistics; head files and every candidate above the "
                           "bound are evaluated in fp64",
                }[mode],
            }
        roof.update({
            "kernel_ms": k_ms,
            "kernel_ms_samples": len(kernel_ms),
            "windows_per_launch": tail_windows,
            "flop_per_window_direct_form": 2.0 * NUM_CH * L,
            "hbm_frac_algorithmic": (alg_bytes / (k_ms * 1e-3) / 1e9) / PEAK_HBM_GBS if k_ms > 0 else 0.0,
        })
        if alt is not None:
            a_name, a_step, a_kms, a_windows, a_flops, a_same, a_geom = alt
            sub = matrix_roofline(a_kms, a_flops) if a_name == "matrix" else spectral_roofline(a_kms, a_windows, a_geom)
            sub.update({"kernel_ms": a_kms, "ms_per_step": a_step * 1e3, "steps": args.other_steps,
                        "windows_per_s": unique_windows / a_step, "same_matches": a_same})
            if a_name == "matrix":
                sub["traffic"] = pmc_traffic_bytes("corr_mfma_kernel") if is_c3 else None
            roof["matrix_cores" if a_name == "matrix" else "spectral"] = sub
        dtype = {"f16": "f16 screening (matrix cores), f64 re-scoring of every returned similarity",
                 "f32": "f32 screening, f64 re-scoring of every returned similarity", "f64": "f64"}[mode]
        if spectral:
            dtype = ("f32 spectral screening (block transforms; spectra kept as 16-bit integers with band scales), "
                     if spectra_q16 else "f32 spectral screening (block transforms), ") + \
                "f64 re-scoring of every returned similarity"
        out = {
            "metric": "candidate windows/sec, 30 s target vs 10 h DB, 13-MFCC; 1/2/4/8 GPU",
            "value": unique_windows / (elapsed / args.steps),
            "unit": "candidate windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_steps,
            "ms_per_step": ms_per_step,
            "cold_ms_per_step": float(np.median(cold_ms)) if cold_ms else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {
                "workload": f"FeatureCorrelation: {args.target_seconds:g} s target ({L} frames) vs "
                            f"{total_files} files x {args.file_seconds:g} s ({frames_per_file} frames each), "
                            f"13 MFCC + loudness, temporalWeight 0.5, top-{args.num_matches}, numPerFile 1, "
                            f"normalize, maxBoost 8",
                "windows": unique_windows,
                "files": total_files,
                "sharding": "by file, round robin, nothing replicated; one all-gather of per-file candidates"
                            if world > 1 else "none",
                "baseline_config": "configs[2]" if (L, args.files_per_gpu) == (2584, 120) else (
                    "configs[4] (per-GPU shard)" if (L, args.files_per_gpu) == (5168, 1500) else "custom"),
            },
            # direct-form work per window grows with the target length: this figure compares runs of different
            # configurations (the spectral kernel does not carry out these operations)
            "whole_job_tflops_direct_form": unique_windows * 2.0 * NUM_CH * L / (elapsed / args.steps) / 1e12,
            "roofline": roof,
            "verified": verified,
        }
        if exact is not None:
            out["roofline"]["exact"] = exact
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(L, frames_per_file, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
    ctx.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if verified else 1


if __name__ == "__main__":
    sys.exit(main())
