"""One rank of the sharded search on a real GPU (started by test_gpu_correlation.py as a child process:
torch has to initialise the GPU before the library does, which a shared pytest process cannot guarantee).
usage: sharded_rank.py PORT MODE RANK WORLD BACKEND [CASES SEED]
MODE: "bounds" (search over bounds), "general" (replicated head files, screen_head off), "headbounds" (replicated
head files and the search over bounds: every rank reports the head files' windows), "exact" (no replicated files,
no screening), "nan" (as "bounds", over a database with constant stretches), "fuzz" (CASES random cases from SEED,
compared with the oracle on every rank), "norms" (two searches with other norms in between, short files first).
Every rank prints the matches (or the fuzz counts) as one JSON line."""
import json
import os
import sys

import torch

torch.cuda.init()
torch.cuda.set_device(0)
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from common import (SHARDED_CASE, STEP, make_plants, norms_default, sharded_case_db, synth_db,  # noqa: E402
                    synth_target)

import strugatzki_amd as S  # noqa: E402

CASE = SHARDED_CASE


def fuzz(port: int, rank: int, world: int, backend: str, cases: int, seed: int) -> int:
    """Random databases and parameters (every rank draws the same ones), all three ways of sharding; every rank
    compares its result with the oracle's.  Prints {"cases", "bad", "skipped"}."""
    import numpy as np
    from common import O, corr_config
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, **kw)
    ctx = S.Context.on_torch_stream(0) if backend == "nccl" else S.Context(0)
    rng = np.random.default_rng(seed)
    bad = skipped = 0
    try:
        for case in range(cases):
            L = int(rng.choice([16, 40, 96, 150]))
            nf = int(rng.integers(1, 11))
            # (one case in three holds files shorter than the target: their ring contents come from other ranks)
            lo = max(1, L // 3) if case % 3 == 2 else L
            frames = [int(rng.integers(lo, 4 * L + 600)) for _ in range(nf)]
            ts = int(rng.integers(0, 20))
            nm, npf = int(rng.integers(1, 6)), int(rng.choice([1, 1, 2, 3]))
            sp = int(rng.choice([0, 0, 10 * STEP]))
            weight = float(rng.choice([0.0, 0.5, 1.0]))
            mode = str(rng.choice(["bounds", "general", "exact", "headbounds"]))
            if npf > 1 or nf < nm:
                mode = "general"
            plants = [q for q in make_plants(int(rng.integers(0, 4)), L, nf, max(frames), ts)
                      if q["file"] < nf and q["offset"] + L <= frames[q["file"]]]
            db = synth_db(nf, frames, plants, seed=int(rng.integers(1, 1 << 30)))
            tgt = synth_target(ts, L)
            want, _, _ = O.correlation_run(corr_config(ts, L, weight=weight, num_matches=nm, num_per_file=npf,
                                                       min_spacing=sp), tgt, norms_default(), db, dense=False)
            nh = min(nm, nf) if mode in ("general", "headbounds") else 0
            head = list(range(nh))
            tail = [g for g in range(nh, nf) if (g - nh) % world == rank]
            ctx.db_reset(db[0].shape[1])
            for g in head + tail:
                ctx.db_add_file(db[g])
            ctx.set_norms(norms_default())
            ctx.corr_set_target(tgt[ts:ts + L], weight, 8.0)
            p = S.SearchParams(nm, npf, sp, STEP, screen=mode != "exact", screen_head=mode in ("bounds", "headbounds"))
            try:
                got = S.search_sharded(ctx, p, head, tail, num_files_total=nf)
            except RuntimeError as e:  # (head files that do not fill the match set: refused on every rank alike)
                if "head files" not in str(e):
                    raise
                skipped += 1
                continue
            ok = [(g.file, g.start, g.stop) for g in got] == [(w.file, w.start, w.stop) for w in want] and all(
                abs(g.sim - w.sim) <= 1e-5 for g, w in zip(got, want))
            if not ok:
                bad += 1
                print(f"rank {rank} case {case} mode {mode} L={L} frames={frames} nm={nm} npf={npf} sp={sp}: "
                      f"{[(g.file, g.start) for g in got]} != {[(w.file, w.start) for w in want]}", file=sys.stderr)
    finally:
        dist.destroy_process_group()
    print(json.dumps({"cases": cases, "bad": bad, "skipped": skipped}))
    ctx.close()
    return 0


def norms_change(port: int, rank: int, world: int, backend: str) -> int:
    """Two searches with a target of the same length and other norms in between, over a database whose first files
    are shorter than the target: the ring positions no file has written normalise to 0, so the raw values the ranks
    hand their short files depend on the norms in force.  Prints both match lists."""
    import numpy as np
    L, ts = 120, 5
    frames = [60, 30, 1000, 119, 700, 400, 260]
    db = synth_db(len(frames), frames)
    tgt = synth_target(ts, L)
    tail = [g for g in range(len(frames)) if g % world == rank]
    ctx = S.Context(0)
    ctx.db_reset(db[0].shape[1])
    for g in tail:
        ctx.db_add_file(db[g])
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, **kw)
    out = []
    try:
        other = norms_default().copy()
        other[:, 0] -= np.float32(0.25)
        other[:, 1] += np.float32(0.5)
        for norms in (norms_default(), other):
            ctx.set_norms(norms)
            ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
            got = S.search_sharded(ctx, S.SearchParams(5, 1, 0, STEP), [], tail, num_files_total=len(frames))
            out.append([[m.file, m.start, m.stop, float(m.sim)] for m in got])
    finally:
        dist.destroy_process_group()
    print(json.dumps(out))
    ctx.close()
    return 0


def main() -> int:
    port, mode, rank, world, backend = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    if mode == "fuzz":
        return fuzz(port, rank, world, backend, int(sys.argv[6]), int(sys.argv[7]))
    if mode == "norms":
        return norms_change(port, rank, world, backend)
    q = CASE
    db = sharded_case_db(mode)
    tgt = synth_target(q["ts"], q["L"])
    if mode in ("general", "headbounds"):  # head files on every rank, the rest dealt round robin
        head = list(range(q["num_matches"]))
        tail = [g for g in range(q["num_matches"], q["nf"]) if (g - q["num_matches"]) % world == rank]
    else:                  # nothing replicated
        head, tail = [], [g for g in range(q["nf"]) if g % world == rank]
    # the RCCL runs use a context on a torch-owned stream (what bench.py does), the gloo runs one with its own
    ctx = S.Context.on_torch_stream(0) if backend == "nccl" else S.Context(0)
    ctx.db_reset(db[0].shape[1])
    for g in head + tail:
        ctx.db_add_file(db[g])
    ctx.set_norms(norms_default())
    ctx.corr_set_target(tgt[q["ts"]:q["ts"] + q["L"]], 0.5, 8.0)
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, **kw)
    try:
        p = S.SearchParams(q["num_matches"], 1, 0, STEP, screen=mode != "exact",
                           screen_head=mode in ("bounds", "headbounds", "nan"))
        got = S.search_sharded(ctx, p, head, tail, num_files_total=q["nf"])
    finally:
        dist.destroy_process_group()
    print(json.dumps([[m.file, m.start, m.stop, float(m.sim)] for m in got]))
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
