#!/usr/bin/env python3
"""Benchmark of the FeatureCorrelation hot path on MI355X.

One step = one complete query against device-resident synthetic feature files:
a NEW target (the timed steps rotate through several distinct targets of the same
length, each with its own planted answers; the raw target frames are resident in
the GPU's memory like the database, as the inputs of the timed region are) is
prepared on the GPU -- normalisation, two-pass statistics, centring, its spectra --
then the sliding correlation of every window of every file, then the reference's
match selection (one all-gather when sharded).  Metric: candidate windows per
second, whole job.  The same step with the target's frames in host memory
(roofline.host_target), the same search with the target left in place
(roofline.same_target), a long leg of the headline step (roofline.steady), the
search with the other screening kernel and the search at reference precision
(fp64, every window) are timed in the same run.

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


def target_start(k: int, L: int) -> int:
    """First frame of target k in the dedicated target file (the targets do not overlap)."""
    return TARGET_START + k * (L + 64)


def plant_list(num: int, L: int, files: list[int], frames_per_file: int, k: int = 0):
    """Copy j of target k goes into files[j] at a distinct offset, noise amplitude 0.02*(j+1)."""
    out = []
    for j in range(num):
        off = (4001 + 1733 * j + 389 * k) % (frames_per_file - L)
        out.append(dict(file=files[j], offset=off, src_file=TARGET_FILE, src_offset=target_start(k, L), length=L,
                        amp=float(np.float32(0.02) * (j + 1)), gain=1.0))
    return out


def norms_default() -> np.ndarray:
    n = np.zeros((NUM_CH, 2), dtype=np.float32)
    n[0] = (0.0, 1.5)
    for c in range(1, NUM_CH):
        n[c] = (np.float32(0.01) * c, np.float32(1.0) + np.float32(0.02) * c)
    return n


def cpu_baseline(L: int, frames_per_file: int, budget_s: float) -> dict:
    """Times the single-threaded CPU oracle, pinned to one core, on a bounded sample of the same workload."""
    from oracle import oracle_lib as O

    allowed = sorted(os.sched_getaffinity(0))
    core = allowed[len(allowed) // 2]
    os.sched_setaffinity(0, {core})
    try:
        out = _cpu_baseline_pinned(O, L, frames_per_file, budget_s)
    finally:
        os.sched_setaffinity(0, set(allowed))
    out["pinned_to_core"] = core
    out["node_cores"] = os.cpu_count()
    out["cores_allowed"] = len(allowed)
    return out


def _cpu_baseline_pinned(O, L: int, frames_per_file: int, budget_s: float) -> dict:
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
    nfiles = max(1, min(64, int(round(want / (frames - L + 1)))))
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
                  f"(C restatement of the reference loop, gcc -O2, one thread pinned to one core)",
    }


PMC_SUMMARY = "r03_pmc_summary.csv"


def finite_json(x):
    """Non-finite floats become null (a JSON document has no NaN)."""
    if isinstance(x, float):
        return x if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: finite_json(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [finite_json(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return finite_json(x.item())
    return x


def pmc_traffic_bytes(kernel: str, summary: str = PMC_SUMMARY):
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
    # a step is about a tenth of a millisecond: the defaults keep the GPU busy for about two seconds (long
    # enough for the clocks to settle and for a once-a-second utilisation sampler to see the run) and still finish
    # at once
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--files-per-gpu", type=int, default=None,
                    help="database files per GPU (default: 120 at N=1, 1500 at N>1)")
    ap.add_argument("--file-seconds", type=float, default=300.0)
    ap.add_argument("--target-seconds", type=float, default=None,
                    help="target length (default: 30 at N=1, 60 at N>1)")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="contexts of the leg with several queries in flight (N = 1; 0 or 1: no such leg)")
    ap.add_argument("--in-flight-steps", type=int, default=12000, help="queries of that leg")
    ap.add_argument("--targets", type=int, default=8,
                    help="distinct targets of the same length the timed steps rotate through (each step sets the next "
                         "one: a search includes the preparation of its target); 1: the target is set once, before "
                         "the timed steps")
    ap.add_argument("--host-targets", action="store_true",
                    help="the timed steps take the targets' raw frames from host memory (default: from the GPU's memory, "
                         "where the inputs of the timed region are resident; the host variant is then timed next to it)")
    ap.add_argument("--steady-steps", type=int, default=12000,
                    help="steps of a second, long leg of the same step as the timed ones (roofline.steady), run when "
                         "--steps is smaller than this; 0 disables")
    ap.add_argument("--same-target-steps", type=int, default=2000,
                    help="steps of the search with the target left in place (roofline.same_target); 0 disables")
    ap.add_argument("--exact-steps", type=int, default=100,
                    help="steps of the reference-precision search (fp64 kernel, every window) timed after the main "
                         "loop at N=1 for roofline.exact; 0 disables")
    ap.add_argument("--num-matches", type=int, default=10)
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed steps for about this long (counted as 0.1 ms each at N = 1, 0.3 ms at N > 1) before the "
                         "warm-up steps, so that the clocks have settled")
    ap.add_argument("--timing-every", type=int, default=32,
                    help="HIP events with one launch in n of the dominant kernel, at most; with few timed steps n is "
                         "lowered so that at least two launches of the timed region carry them.  A launch that "
                         "carries events costs the host about 20 us more than one that does not (measured: "
                         "tools/search_loop.cpp), so the kernel's average duration over the timed region comes from "
                         "a sample of its launches")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="CPU baseline budget; 0 disables")
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
    rccl_ranks = None
    if sharded:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        rccl_ranks = dist.get_world_size()

    sr = 44100.0
    L = int((int(args.target_seconds * sr + 0.5) + TARGET_START * STEP + STEP // 2) // STEP) - TARGET_START
    frames_per_file = -(-int(args.file_seconds * sr) // STEP) - 1  # ceil(S/step) - 1
    head_n = args.num_matches
    # weak scaling: every rank holds files_per_gpu files of the database, dealt round robin
    total_files = world * args.files_per_gpu
    # Targets and their planted answers: copy j of target k sits in file spread[j] + k (every file holds at most one
    # copy), noise growing with j, so the expected result of a search with target k is its own ten copies in order.
    n_tgt = max(1, args.targets)
    while n_tgt > 1 and total_files < args.num_matches * n_tgt + 1:
        n_tgt -= 1
    spread = [int(round(j * (total_files - n_tgt) / max(args.num_matches - 1, 1))) for j in range(args.num_matches)]
    plants_of = [plant_list(args.num_matches, L, [f + k for f in spread], frames_per_file, k) for k in range(n_tgt)]
    by_file = {}
    for plants in plants_of:
        for p in plants:
            by_file.setdefault(p["file"], []).append(
                N.Plant(p["file"], p["offset"], p["src_file"], p["src_offset"], p["length"], p["amp"], p["gain"]))

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
        ctx.db_add_synthetic(SEED, g, frames_per_file, by_file.get(g, []))
    ctx.set_norms(norms_default())
    # the targets: raw frames in host memory, as a caller holds them
    ctx_t = S.Context(local_rank)
    ctx_t.db_reset(NUM_CH)
    ctx_t.db_add_synthetic(SEED, TARGET_FILE, target_start(n_tgt, L) + 16, [])
    tfile = ctx_t.db_read_file(0)
    targets = [np.ascontiguousarray(tfile[target_start(k, L):target_start(k, L) + L]) for k in range(n_tgt)]
    ctx_t.close()
    # ... and in the GPU's memory, where the timed steps take them from: the inputs of the timed region are resident
    # in HBM when it starts (a query's feature frames come out of the extraction on the device, or a service keeps
    # its targets there); the same step with the frames in host memory is reported next to it (roofline.host_target)
    targets_dev = None
    if not args.host_targets:
        targets_dev = [torch.from_numpy(t).to(f"cuda:{local_rank}") for t in targets]
        torch.cuda.synchronize()

    def set_target(k, host=False):
        if host or targets_dev is None:
            ctx.corr_set_target(targets[k], 0.5, 8.0)
        else:
            ctx.corr_set_target_device(targets_dev[k].data_ptr(), L, 0.5, 8.0)

    ctx.corr_set_target(targets[0], 0.5, 8.0)
    ctx.set_screen_kernel(args.screen_kernel)

    params = S.SearchParams(args.num_matches, 1, 0, STEP, capacity=1 << 16, screen=not args.exact_only,
                            screen_mode=args.screen_mode)
    mode = "f16" if args.exact_only is False and args.screen_mode != "f32" and ctx.screen_f16_available else (
        "f64" if args.exact_only else "f32")
    unique_windows = total_files * (frames_per_file - L + 1)

    def search_only():
        if sharded:
            return S.search_sharded(ctx, params, head_global, tail_global, num_files_total=total_files)
        return S.search(ctx, params)

    counter = [0]
    last_of = [None] * n_tgt

    def step():
        """The headline step: the next target of the rotation, prepared on the GPU, then the search."""
        k = counter[0] % n_tgt
        counter[0] += 1
        if n_tgt > 1:
            set_target(k)
        m = search_only()
        last_of[k] = m
        return m

    def step_host_target():
        """The same with the target's raw frames in host memory (copied into pinned memory, read across the bus)."""
        k = counter[0] % n_tgt
        counter[0] += 1
        set_target(k, host=True)
        return search_only()

    def sync():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def keys(ms):
        return [(m.file, m.start // STEP) for m in ms]

    # Host side hygiene of a long-running search service: the interpreter's full
    # garbage collections walk every object torch created at import (tens of ms);
    # park those objects in the permanent generation before the loop.
    import gc
    gc.collect()
    gc.freeze()

    # HIP events travel with one launch in timing_every of the dominant kernel: at most --timing-every, and few
    # enough that two launches of the timed region carry them (such a launch costs the host 20 us more)
    def every_for(steps):
        return max(1, min(args.timing_every, steps // 2)) if mode == "f16" else 1

    def timed_leg(fn, steps):
        """`steps` calls of fn between two synchronisations: (seconds, kernel times of the sampled launches)."""
        te = every_for(steps)
        ctx.set_kernel_timing(te)
        kms = []
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
            if mode != "f16":  # several runs per step: the tail launch is the last one
                kms.append(ctx.corr_last_kernel_ms())
        sync()
        dt = time.perf_counter() - t0
        # the events are read here, after the timed region (a host-side wait on them between steps would sit inside it)
        if mode == "f16":  # one run per step
            kms = [float(x) for x in ctx.kernel_ms_history(min(256, -(-steps // te))) if x > 0]
        return dt, kms

    # Clock settling, untimed and reported as such: after an idle period the chip needs some tens of
    # milliseconds of load before it holds its clocks; the first steps then run 5-10 % slower than the rest.
    # (a fixed count, the same on every rank: the sharded step holds a collective)
    # (the untimed steps carry timing events like the timed ones: the library creates its events on first use)
    ctx.set_kernel_timing(every_for(args.steps))
    matches = step()
    # which kernel the layout of this database and target uses for the bounds
    spectral = mode == "f16" and ctx.screen_kernel_in_use == "spectral"
    fft_n, fft_hop, fft_blocks = ctx.screen_geometry
    spectra_q16 = ctx.spectra_format_in_use == "q16"
    settle_steps = int(args.settle_ms / (0.3 if world > 1 else 0.1))  # (a step: about 0.1 ms at N = 1, 0.57 at N > 1)
    for _ in range(settle_steps):
        step()
    for _ in range(args.warmup):
        step()
    for k in range(n_tgt):
        last_of[k] = None
    elapsed, kernel_ms = timed_leg(step, args.steps)
    rank_elapsed = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        ts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        rank_elapsed = [float(x.item()) for x in ts]
        elapsed = max(rank_elapsed)
    screened_work = ctx.corr_last_work()
    # known answers: the matches of a search with target k are its planted copies, in noise order, at their offsets
    verified_targets = sum(1 for k in range(n_tgt) if last_of[k] is not None and
                           keys(last_of[k]) == [(p["file"], p["offset"]) for p in plants_of[k]])
    searched_targets = sum(1 for k in range(n_tgt) if last_of[k] is not None)
    verified = searched_targets > 0 and verified_targets == searched_targets

    # The same step for longer (at least a second of GPU work): the figure a few dozen timed steps cannot give
    # reliably, and long enough for a utilisation sampler to see the kernel.
    steady = None
    if 0 < args.steps < args.steady_steps:
        s_elapsed, s_kms = timed_leg(step, args.steady_steps)
        if world > 1:
            t = torch.tensor([s_elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            s_elapsed = float(t.item())
        steady = (s_elapsed / args.steady_steps, s_kms)

    # The search alone, the target left in place (what the timed steps of earlier versions of this benchmark were).
    same = None
    if args.same_target_steps > 0 and n_tgt > 1:
        ctx.corr_set_target(targets[0], 0.5, 8.0)
        search_only()
        q_elapsed, q_kms = timed_leg(search_only, args.same_target_steps)
        if world > 1:
            t = torch.tensor([q_elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            q_elapsed = float(t.item())
        same = (q_elapsed / args.same_target_steps, q_kms)

    # The headline step with the target's frames in host memory (the boundary also takes host buffers: that rate is
    # reported, it is not `value`).
    host_leg = None
    if args.same_target_steps > 0 and n_tgt > 1 and targets_dev is not None:
        for _ in range(20):
            step_host_target()
        h_elapsed, h_kms = timed_leg(step_host_target, args.same_target_steps)
        if world > 1:
            t = torch.tensor([h_elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            h_elapsed = float(t.item())
        host_leg = (h_elapsed / args.same_target_steps, h_kms)

    # Several queries in flight (sgz_corr_search_batch): the headline step dealt over replica contexts, a host thread
    # each.  A query alone is a chain of dependent launches that leaves the chip idle for about half of its
    # duration; the kernels of several queries fill each other's gaps.  Reported next to the headline, not as it:
    # `value` stays the rate of queries answered one after another.
    in_flight = None
    if not sharded and world == 1 and args.in_flight > 1 and args.in_flight_steps > 0 and mode == "f16" \
            and targets_dev is not None:
        # (a secondary leg: whatever goes wrong in it is reported in its place and leaves the line's other figures alone)
        replicas = [ctx]
        try:
            for _ in range(args.in_flight - 1):
                r = S.Context(local_rank)
                r.db_reset(NUM_CH)
                for g in head_global + tail_global:
                    r.db_add_synthetic(SEED, g, frames_per_file, by_file.get(g, []))
                r.set_norms(norms_default())
                r.set_screen_kernel(args.screen_kernel)
                replicas.append(r)
            nq = args.in_flight_steps
            te_b = max(1, min(args.timing_every, nq // (2 * args.in_flight)))  # (events on the first context's launches)
            ctx.set_kernel_timing(te_b)
            ptrs = [targets_dev[q % n_tgt].data_ptr() for q in range(nq)]
            S.search_batch(replicas, params, ptrs[:min(nq, 2000)], 0.5, 8.0, target_frames=L)  # layouts built, clocks held
            sync()
            tb = time.perf_counter()
            res = S.search_batch(replicas, params, ptrs, 0.5, 8.0, target_frames=L)
            b_elapsed = time.perf_counter() - tb
            b_kms = [float(x) for x in ctx.kernel_ms_history(min(256, nq // (args.in_flight * te_b))) if x > 0]
            ok = sum(1 for q in range(nq) if keys(res[q]) == [(p["file"], p["offset"]) for p in plants_of[q % n_tgt]])
            in_flight = {
                "what": "the headline step (a new target per search) with several queries in flight: replica contexts "
                        "(each with its own copy of the database, stream and target state), a host thread per context, "
                        "one sgz_corr_search_batch call over all queries",
                "contexts": args.in_flight,
                "queries": nq,
                "ms_per_query": b_elapsed / nq * 1e3,
                "windows_per_s": unique_windows / (b_elapsed / nq),
                # (start to end of the first context's launches: the kernels of the other queries share the chip)
                "kernel_ms": float(np.mean(b_kms)) if b_kms else None,
                "kernel_ms_samples": len(b_kms),
                "queries_verified": ok,
                "verified": ok == nq,
            }
        except Exception as e:  # noqa: BLE001
            in_flight = {"error": f"{type(e).__name__}: {e}"}
        for r in replicas[1:]:
            r.close()
        ctx.corr_set_target(targets[0], 0.5, 8.0)

    # The collective alone (sharded runs): the all-gather of the candidate blocks, their copy to pinned memory and
    # the event behind it, as the search queues them, without the kernels in front.
    collective_ms = None
    if sharded:
        from strugatzki_amd.search import exchange_only
        for _ in range(20):
            exchange_only(ctx, params)
        sync()
        tx = time.perf_counter()
        for _ in range(200):
            exchange_only(ctx, params)
        sync()
        collective_ms = (time.perf_counter() - tx) / 200 * 1e3

    ctx.corr_set_target(targets[0], 0.5, 8.0)
    matches = search_only()

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
            ctx.synchronize()
            tc = time.perf_counter()
            ctx.corr_set_target(targets[0], 0.5, 8.0)
            search_only()
            cold_ms.append((time.perf_counter() - tc) * 1e3)

    # The same search with the other screening kernel, timed in the same run (the layout is rebuilt, so the first
    # steps are cold and untimed).
    alt = None
    if world == 1 and not sharded and mode == "f16" and args.other_steps > 0 and args.screen_kernel == "auto":
        other_name = "matrix" if spectral else "spectral"
        ctx.set_screen_kernel(other_name)
        for _ in range(100):
            om = search_only()
        if ctx.screen_kernel_in_use == other_name:
            o_elapsed, o_ms = timed_leg(search_only, args.other_steps)
            om = search_only()
            o_windows, o_flops = ctx.corr_last_work()
            alt = (other_name, o_elapsed / args.other_steps, float(np.mean(o_ms)) if o_ms else 0.0, o_windows, o_flops,
                     [(m.file, m.start, m.stop, m.sim) for m in om] == [(m.file, m.start, m.stop, m.sim) for m in matches],
                     ctx.screen_geometry)
        ctx.set_screen_kernel(args.screen_kernel)
        search_only()

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
            "verified": keys(em) == [(p["file"], p["offset"]) for p in plants_of[0]],
            "same_matches_as_screened": [(m.file, m.start, m.stop) for m in em] == [(m.file, m.start, m.stop) for m in matches],
            "note": "every window evaluated by the fp64 kernel (v_fmac_f64, fp64 vector peak 78.6 TFLOP/s), selection "
                    "replayed over its values: the search at the reference's arithmetic, no screening",
        }

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # dominant kernel: the launch that evaluates every window of the step
        tail_windows, tail_flops = screened_work
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else 0.0
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
            # what the kernel is built to move: the blocks' spectra and 8 B of cached statistics per window in (20 B in
            # blocks where some window's boost comes near maxBoost: none in this workload); out only the list of the
            # few windows that can hold their file's maximum and one bound per file
            q16 = spectra_q16
            # (16-bit spectra: 4 bytes per complex bin and a 4-byte scale per band of 64 bins; every workgroup also
            # reads the target's spectra, which the caches serve)
            moved = blocks * (2.0 * NUM_CH * n + NUM_CH * n / 32.0 if q16 else 4.0 * NUM_CH * n) + windows * 8.0
            return {
                "bound": "hbm", "kernel": "corr_fft_kernel", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ach / PEAK_HBM_GBS,
                "algorithmic_bytes": windows * 64.0,
                "bytes_by_design": moved,
                "gbs_by_design": moved / (kms * 1e-3) / 1e9 if kms > 0 else 0.0,
                "frac_by_design": moved / (kms * 1e-3) / 1e9 / PEAK_HBM_GBS if kms > 0 else 0.0,
                "transform_length": n, "windows_per_block": hop, "blocks": blocks,
                "spectra": "16-bit integers, one scale per 64 bins" if q16 else "fp32",
                "note": "one launch over all files, one workgroup per block of N frames: the cross terms of a block's "
                        "N - L + 1 windows are one circular correlation, evaluated from spectra of the blocks kept "
                        "in HBM with the database (16-bit integers with a scale per band, whose measured distance to "
                        "the fp32 spectra is part of the certified error; product with the target's spectra, channels "
                        "summed in the frequency domain, one fp32 inverse transform in LDS); achieved = 64 algorithmic bytes per window (one new "
                        "frame in, similarity and boost out) over the kernel's duration, against 8 TB/s; "
                        "bytes_by_design is what this formulation has to move (overlapping blocks, cached window "
                        "statistics; the search reads candidate lists, not an upper bound per window), frac_by_design "
                        "that over the same duration and peak.  It yields certified bounds, and every "
                        "window the bounds cannot exclude is re-evaluated in fp64 from the raw frames before the "
                        "selection replay",
            }

        def traffic_of(roof, kernel):
            """HBM bytes per launch from the committed counter passes of this command (not measured in this run)."""
            t = pmc_traffic_bytes(kernel) if is_c3 else None
            roof["traffic"] = t
            roof["traffic_source"] = f"profiles/{PMC_SUMMARY} (rocprofv3 --pmc passes of this command, committed)" \
                if t is not None else None
            if t is not None and roof.get("kernel_ms"):
                roof["frac_by_traffic"] = t / (roof["kernel_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS

        is_c3 = (L, total_files) == (2584, 120)
        if spectral:
            roof = spectral_roofline(k_ms, tail_windows, (fft_n, fft_hop, fft_blocks))
        elif mode == "f16":
            roof = matrix_roofline(k_ms, tail_flops)
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
            "kernel_ms": k_ms if kernel_ms else None,
            "kernel_ms_samples": len(kernel_ms),
            "windows_per_launch": tail_windows,
            "flop_per_window_direct_form": 2.0 * NUM_CH * L,
            "hbm_frac_algorithmic": (alg_bytes / (k_ms * 1e-3) / 1e9) / PEAK_HBM_GBS if k_ms > 0 else 0.0,
        })
        if mode == "f16":
            traffic_of(roof, "corr_fft_kernel" if spectral else "corr_mfma_kernel")

        def leg(per_step, kms, steps, what):
            km = float(np.mean(kms)) if kms else 0.0
            sub = spectral_roofline(km, tail_windows, (fft_n, fft_hop, fft_blocks)) if spectral else (
                matrix_roofline(km, tail_flops) if mode == "f16" else {})
            sub.pop("note", None)
            sub.update({"what": what, "steps": steps, "ms_per_step": per_step * 1e3,
                        "windows_per_s": unique_windows / per_step, "kernel_ms": km if kms else None,
                        "kernel_ms_samples": len(kms)})
            return sub

        if steady is not None:
            roof["steady"] = leg(steady[0], steady[1], args.steady_steps,
                                 "the timed step (a new target per search) for longer, after the timed steps")
        if same is not None:
            roof["same_target"] = leg(same[0], same[1], args.same_target_steps,
                                      "the search alone, the target left in place (no target preparation in the step)")
        if host_leg is not None:
            roof["host_target"] = leg(host_leg[0], host_leg[1], args.same_target_steps,
                                      "the timed step with the target's raw frames in host memory (copied to pinned "
                                      "memory, read across the bus by the first kernel)")
        if alt is not None:
            a_name, a_step, a_kms, a_windows, a_flops, a_same, a_geom = alt
            sub = matrix_roofline(a_kms, a_flops) if a_name == "matrix" else spectral_roofline(a_kms, a_windows, a_geom)
            sub.update({"kernel_ms": a_kms, "ms_per_step": a_step * 1e3, "steps": args.other_steps,
                        "windows_per_s": unique_windows / a_step, "same_matches": a_same,
                        "what": "the search alone (target left in place) with the other screening kernel"})
            if a_name == "matrix":
                traffic_of(sub, "corr_mfma_kernel")
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
            "step": ("a new target per search: one of %d distinct targets in rotation, raw frames in %s memory -> "
                     "normalisation, statistics, centring and spectra on the GPU, then the search"
                     % (n_tgt, "host" if targets_dev is None else "GPU")) if n_tgt > 1
                    else "the search alone, the target set once before the timed steps",
            "targets": n_tgt,
            "targets_verified": verified_targets,
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
        if sharded:
            out["rccl_ranks"] = rccl_ranks
            out["rank_ms_per_step"] = {"min": min(rank_elapsed) / args.steps * 1e3,
                                       "max": max(rank_elapsed) / args.steps * 1e3}
            out["collective_ms_per_step"] = collective_ms
            out["collective_share_of_step"] = collective_ms / ms_per_step if collective_ms is not None else None
        if in_flight is not None:
            out["queries_in_flight"] = in_flight
        if exact is not None:
            out["roofline"]["exact"] = exact
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(L, frames_per_file, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(finite_json(out), allow_nan=False), flush=True)
    ctx.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if verified else 1


if __name__ == "__main__":
    sys.exit(main())
