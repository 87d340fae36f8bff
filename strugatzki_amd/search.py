"""Database search = GPU similarity + exact replay of the reference's match selection.

The reference walks the database files one after another and keeps two
priority sets whose gate depends on what earlier files produced
(impl/FeatureCorrelationImpl.scala:113-150, :234, :399-400).  The GPU computes
the similarity of every window of every file at once; the selection is then
replayed on the host in file order:

  1. "head": files are fed densely until the global set holds num_matches
     entries.  From then on a window can only be offered if its sim exceeds
     the sim of the worst global entry, and that bound never falls.
  2. "tail": for all remaining files only windows above that bound are fetched
     from the GPU (stream compaction) and fed in file / offset order.  By default
     the tail similarities come from the screening kernel (fp32 products, about
     1e-7 off); windows within `margin` of the bound are re-scored exactly in
     fp64 on the GPU before they are fed, so the result is the one the exact
     kernel gives.

Sharded over several GPUs (one process each), every rank holds the head files
plus its own slice of the tail; the ranks exchange their tail candidates with
one all-gather and then run the same replay, so every rank ends with the same
list as a single-GPU run.
"""
from __future__ import annotations

from dataclasses import dataclass
import ctypes as C
from typing import Sequence

import numpy as np

from . import _native as N
from .engine import Context, Selector, iter_files


@dataclass
class SearchParams:
    num_matches: int = 1
    num_per_file: int = 1
    min_spacing: int = 0      # sample frames
    step_size: int = 512      # fftSize / fftOverlap
    capacity: int = 1 << 16   # candidate records fetched per rank
    screen: bool = True       # tail files: screening kernel + exact re-scoring of the candidates
    screen_mode: str = "auto"  # "f16": half-precision matrix-core kernel, "f32": fp32-product kernel,
                               # "auto": f16 where available (normalised features, target fits LDS), else f32
    margin: float | None = None  # screening keeps windows with sim > T0 - margin; None: per mode (MARGIN)
    screen_head: bool = True   # numPerFile == 1 and f16 screening: search over bounds, no exact head run


# Default margins.  f32: many times the largest screening error observed against the exact kernel
# (1.2e-7).  f16: that kernel already writes an upper bound of the similarity, the margin only covers the
# rounding of that sum.
MARGIN = {"f32": 1e-5, "f16": 1e-6}


def _screen_mode(ctx: Context, p: SearchParams) -> str:
    if p.screen_mode == "auto":
        return "f16" if ctx.screen_f16_available else "f32"
    if p.screen_mode not in MARGIN:
        raise ValueError(f"unknown screen_mode {p.screen_mode!r}")
    return p.screen_mode


def _run_screen(ctx: Context, mode: str, fb: int, fe: int) -> None:
    (ctx.corr_run_screen_f16 if mode == "f16" else ctx.corr_run_screen)(fb, fe)


def _feed_dense_head(ctx: Context, sel: Selector, files: Sequence[int], global_ids: Sequence[int]) -> int:
    """Feeds all given files densely, in order (dense feeding is exact whether or
    not the set is already full). Returns how many were fed."""
    files = list(files)
    if not files:
        return 0
    dense = ctx.corr_read_dense_range(files[0], files[-1] + 1)  # `files` is a consecutive range
    for (sim, boost), gid in zip(dense, global_ids):
        sel.feed_dense(gid, sim, boost)
    return len(files)


_SCREEN_CODES = {"f32": N.SGZ_SCREEN_F32, "f16": N.SGZ_SCREEN_F16, "auto": N.SGZ_SCREEN_AUTO}


def search(ctx: Context, p: SearchParams) -> list[N.Match]:
    """Single-GPU search over all files of the context, in their index order: one native call
    (sgz_corr_search) that runs the kernels and replays the selection."""
    if p.screen_mode not in _SCREEN_CODES:
        raise ValueError(f"unknown screen_mode {p.screen_mode!r}")
    mode = _SCREEN_CODES[p.screen_mode] if p.screen else N.SGZ_SCREEN_NONE
    return ctx.corr_search(p.num_matches, p.num_per_file, p.min_spacing, p.step_size, mode,
                           over_bounds=p.screen and p.screen_head, capacity=p.capacity,
                           margin=-1.0 if p.margin is None else p.margin)


def search_batch(contexts: Sequence[Context], p: SearchParams, targets, temporal_weight: float, max_boost: float,
                 target_frames: int | None = None) -> list[list[N.Match]]:
    """Several queries in flight on one GPU (sgz_corr_search_batch): `contexts` are replicas (the same files in
    the same order, the same norms), `targets` either arrays [frames][channels] in host memory or addresses (int)
    of such frames in the GPU's memory (then `target_frames` says how many).  A query is corr_set_target + search();
    the result of query q is what search() on one context returns for targets[q]."""
    if p.screen_mode not in _SCREEN_CODES:
        raise ValueError(f"unknown screen_mode {p.screen_mode!r}")
    mode = _SCREEN_CODES[p.screen_mode] if p.screen else N.SGZ_SCREEN_NONE
    nq = len(targets)
    if nq == 0:
        return []
    on_device = isinstance(targets[0], int)
    if on_device:
        if target_frames is None:
            raise ValueError("target_frames is needed with targets in GPU memory")
        ptrs = (C.c_void_p * nq)(*targets)
        keep = None
    else:
        keep = [np.ascontiguousarray(t, dtype=np.float32) for t in targets]
        target_frames = keep[0].shape[0]
        if any(k.shape != keep[0].shape for k in keep):
            raise ValueError("the targets of a batch have the same length")
        ptrs = (C.c_void_p * nq)(*[k.ctypes.data for k in keep])
    n = len(contexts)
    handles = (C.c_void_p * n)(*[c._h for c in contexts])
    cap = max(p.num_matches, 1)
    out = (N.Match * (cap * nq))()
    counts = (C.c_int * nq)()
    rc = N.lib.sgz_corr_search_batch(handles, n, ptrs, 1 if on_device else 0, nq, target_frames, temporal_weight,
                                     max_boost, p.num_matches, p.num_per_file, p.min_spacing, p.step_size, mode,
                                     1 if (p.screen and p.screen_head) else 0, p.capacity,
                                     -1.0 if p.margin is None else p.margin, out, cap, counts)
    for c in contexts:
        c.target_frames = target_frames
    if rc != N.SGZ_OK:
        raise N.NativeError(rc, "; ".join(N.lib.sgz_last_error(c._h).decode() for c in contexts))
    return [[out[q * cap + i] for i in range(counts[q])] for q in range(nq)]


def search_punch(ctx: Context, p: SearchParams, target_in: np.ndarray, weight_in: float, target_out: np.ndarray,
                 weight_out: float, max_boost: float, min_punch: int, max_punch: int) -> list[N.Match]:
    """Punch-in / punch-out search (impl/FeatureCorrelationImpl.scala:182-397) over the files of the context.
    `target_in` / `target_out`: raw frames of the two punch spans; `min_punch` / `max_punch` in feature
    frames.  Both targets are correlated with every window of every file by the exact kernel; the pairing
    is the reference's sequential loop (Selector.feed_file_punch).  Leaves the punch-out target set."""
    nf = ctx.num_files
    sel = Selector(p.num_matches, p.num_per_file, p.min_spacing, p.step_size, len(target_in))
    try:
        dense = []
        # the punch-in pass stops min_punch frames before the end of a file (:182-185)
        for tgt, w, trim in ((target_in, weight_in, min_punch), (target_out, weight_out, 0)):
            ctx.corr_set_tail_trim(trim)
            ctx.corr_set_target(tgt, w, max_boost)
            ctx.corr_run(0, nf)
            dense.append(ctx.corr_read_dense_range(0, nf) if nf else [])
        for f in range(nf):
            (isim, iboost), (osim, oboost) = dense[0][f], dense[1][f]
            sel.feed_file_punch(f, ctx.file_frames(f), len(target_in), len(target_out), min_punch, max_punch,
                                isim, iboost, osim, oboost)
        return sel.result()
    finally:
        ctx.corr_set_tail_trim(0)
        sel.close()


def shard_files(num_files: int, head: int, rank: int, world: int) -> list[int]:
    """Global ids of the tail files rank `rank` owns (round robin keeps equal-length files balanced)."""
    return [g for g in range(head, num_files) if (g - head) % world == rank]


def exchange_candidates(rec, cnt, capacity: int, group=None) -> np.ndarray | None:
    """The one collective of the search: all-gather of fixed-size candidate
    buffers and their fill counts.  `rec` is an int32 [capacity][4] tensor whose
    rows are sgz_candidate records (file already translated to the global id),
    `cnt` a 1-element int64 tensor; both live on the GPU for the nccl (RCCL)
    backend and on the host for gloo.  Returns all ranks' candidates sorted by
    (file, offset), records that several ranks hold (replicated files) once;
    None when some rank's buffer overflowed (every rank sees that)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if rec.is_cuda and dist.get_backend(group) == "gloo":  # gloo gathers host tensors only
        rec, cnt = rec.cpu(), cnt.cpu()
    recs = [torch.empty_like(rec) for _ in range(world)]
    cnts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    if max(counts) > capacity:
        return None
    dist.all_gather(recs, rec, group=group)
    parts = [recs[r][: counts[r]].cpu().numpy() for r in range(world)]
    allc = np.concatenate(parts, axis=0) if parts else np.zeros((0, 4), np.int32)
    cands = np.ascontiguousarray(allc, dtype=np.int32).view(N.CANDIDATE_DTYPE).reshape(-1)
    cands = _sort_candidates(cands)
    return cands


def finish_selection(sel: Selector, cands: np.ndarray) -> list[N.Match]:
    """Tail replay over (file, offset)-sorted candidates, then the result list."""
    sel.feed_candidate_list(cands)
    return sel.result()


def _device_buffers(cap: int):
    import torch

    dev = torch.device("cuda", torch.cuda.current_device())
    rec = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.current_stream().synchronize()  # buffers are zeroed before the library's stream writes them
    return rec, cnt, dev


def _to_global_ids(rec, cnt, cap: int, local_to_global: Sequence[int], dev) -> None:
    import torch

    n_local = min(int(cnt.item()), cap)
    if n_local:
        lut = torch.tensor(list(local_to_global), dtype=torch.int32, device=dev)
        rec[:n_local, 0] = lut[rec[:n_local, 0].long()]


_EXCHANGE_ROWS = 1023  # candidate rows that travel in the first (usually only) all-gather, plus one header row:
# 16 KiB per rank (a search over bounds yields 2-3 records per file)


def _sort_candidates(cands: np.ndarray) -> np.ndarray:
    """(file, offset) order, records that several ranks hold (replicated files) once."""
    if len(cands) > 1:
        key = (cands["file"].astype(np.int64) << 32) | cands["offset"].astype(np.int64)
        order = np.argsort(key, kind="stable")
        cands, key = cands[order], key[order]
        keep = np.ones(len(cands), dtype=bool)
        keep[1:] = key[1:] != key[:-1]
        if not keep.all():
            cands = cands[keep]
    return cands


def _rank_maps(ctx: Context, local_to_global: Sequence[int], group) -> dict:
    """local index -> global file id of every rank (gathered once per placement, kept on the context until its
    db_reset: the gathering is a collective, so when one rank's files change every rank has to start a new
    database, or some ranks would enter it and others not)."""
    import torch
    import torch.distributed as dist

    cache = ctx.__dict__.setdefault("_rank_maps", {})
    # (every search asks: the tuple search_sharded keeps per placement is recognised by identity, without hashing
    # its thousands of entries)
    last = cache.get("_last")
    if last is not None and last[0] is local_to_global and last[1] == id(group):
        return last[2]
    key = (tuple(local_to_global), id(group))
    if key not in cache:
        world = dist.get_world_size(group)
        on_gpu = dist.get_backend(group) != "gloo"
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        n = torch.tensor([len(local_to_global)], dtype=torch.int64, device=dev)
        ns = [torch.empty_like(n) for _ in range(world)]
        dist.all_gather(ns, n, group=group)
        width = max(int(x.item()) for x in ns)
        mine = torch.full((max(width, 1),), -1, dtype=torch.int32, device=dev)
        mine[: len(local_to_global)] = torch.tensor(list(local_to_global), dtype=torch.int32, device=dev)
        maps = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(maps, mine, group=group)
        per_rank = [m.cpu().numpy()[: int(k.item())] for m, k in zip(maps, ns)]
        offsets = np.zeros(world + 1, dtype=np.int64)
        offsets[1:] = np.cumsum([len(m) for m in per_rank])
        flat = np.ascontiguousarray(np.concatenate(per_rank) if per_rank else np.zeros(0), dtype=np.int32)
        # frame counts of every rank's files (for the ring contents of files shorter than the target)
        fr = torch.zeros((max(width, 1),), dtype=torch.int64, device=dev)
        fr[: len(local_to_global)] = torch.tensor([ctx.file_frames(i) for i in range(len(local_to_global))],
                                                  dtype=torch.int64, device=dev)
        frs = [torch.empty_like(fr) for _ in range(world)]
        dist.all_gather(frs, fr, group=group)
        frames = {}
        owner = {}
        for r, (m, f) in enumerate(zip(per_rank, frs)):
            f = f.cpu().numpy()
            for i, g in enumerate(m):
                frames.setdefault(int(g), int(f[i]))  # (replicated files: the same everywhere)
                owner.setdefault(int(g), (r, i))
        cache.clear()
        cache[key] = dict(per_rank=per_rank, flat=flat, offsets=offsets, frames=frames, owner=owner, ring_for=None)
    cache["_last"] = (local_to_global, id(group), cache[key])
    return cache[key]


def _exchange_ring_images(ctx: Context, local_to_global: Sequence[int], group) -> None:
    """Files shorter than the target are scored over their own frames followed by what earlier files of the
    reference's single loop left in its ring (impl/FeatureCorrelationImpl.scala:154, 190-195).  With the files
    spread over ranks those frames may live elsewhere: every rank works out the same list of source runs
    (ring_runs over the frame counts of all files in global order), the ranks all-gather the runs' frames
    (each rank fills in the runs it holds), and every rank hands its short files their ring contents
    (Context.corr_set_ring_image).  One collective, and only when the database holds such a file; done once per
    (placement, target length, norms): ring positions no file has written hold raw values that normalise to 0, and
    those depend on the norms in force."""
    import torch
    import torch.distributed as dist

    from .engine import ring_runs

    maps = _rank_maps(ctx, local_to_global, group)
    L = ctx.target_frames
    done_for = (L, ctx.__dict__.get("_norms_key"))
    if maps["ring_for"] == done_for:
        return
    frames = maps["frames"]
    nfiles = len(frames)
    if sorted(frames) != list(range(nfiles)):
        maps["ring_for"] = done_for
        return  # not a dense global numbering: the caller's own affair
    counts = [frames[g] for g in range(nfiles)]
    if not any(0 < c < L for c in counts):
        maps["ring_for"] = done_for
        return
    runs = ring_runs(counts, L)
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    nch = ctx.num_channels
    total = int(runs["count"].sum())
    mine = np.zeros((max(total, 1), nch), dtype=np.float32)
    pos = 0
    spans = []
    for r in runs:
        n = int(r["count"])
        spans.append((pos, n))
        src = int(r["src_file"])
        if src >= 0 and maps["owner"][src][0] == rank:
            mine[pos:pos + n] = ctx.db_read_frames(maps["owner"][src][1], int(r["src_frame"]), n)
        pos += n
    on_gpu = dist.get_backend(group) != "gloo"
    t = torch.from_numpy(mine)
    if on_gpu:
        t = t.to(torch.device("cuda", torch.cuda.current_device()))
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t, group=group)
    parts = [q.cpu().numpy() for q in parts]
    zero = ctx.norm_zero_frame()
    local_of = {int(g): i for i, g in enumerate(local_to_global)}
    k = 0
    while k < len(runs):
        g = int(runs[k]["file"])
        own = counts[g]
        img = np.zeros((L - own, nch), dtype=np.float32)
        while k < len(runs) and int(runs[k]["file"]) == g:
            r = runs[k]
            p0, n = spans[k]
            src = int(r["src_file"])
            d = int(r["dest"]) - own
            img[d:d + n] = zero if src < 0 else parts[maps["owner"][src][0]][p0:p0 + n]
            k += 1
        if g in local_of:
            ctx.corr_set_ring_image(local_of[g], img)
    maps["ring_for"] = done_for  # (only now: an exception on the way leaves the exchange to be done again)


def _exchange_buffers(ctx: Context, cap: int, world: int, on_gpu: bool):
    """[header + cap records] of this rank and the gather target, kept on the context between searches."""
    import torch

    key = (cap, world, on_gpu)
    cache = ctx.__dict__.setdefault("_exchange_buffers", {})
    if key not in cache:
        dev = torch.device("cuda", torch.cuda.current_device())
        rec = torch.zeros((cap + 1, 4), dtype=torch.int32, device=dev)
        rows = min(cap, _EXCHANGE_ROWS) + 1
        gathered = torch.zeros((world * rows, 4), dtype=torch.int32, device=dev if on_gpu else torch.device("cpu"))
        host = torch.zeros((world * rows, 4), dtype=torch.int32).pin_memory() if on_gpu else gathered
        torch.cuda.current_stream().synchronize()
        cache.clear()
        cache[key] = dict(rec=rec, send=rec[:rows], gathered=gathered, host=host, rows=rows,
                          count_ptr=rec.data_ptr(), rec_ptr=rec[1:].data_ptr(),
                          view=host.numpy().reshape(world, rows, 4), event=torch.cuda.Event())
    return cache[key]


def exchange_only(ctx: Context, p: SearchParams, group=None) -> None:
    """The collective of a sharded search by itself, as _sharded_over_bounds queues it (all-gather of the candidate
    blocks, their copy to pinned memory, the event the host polls) over whatever the buffers hold: for timing the
    exchange apart from the kernels.  Needs an earlier sharded search on this context (its buffers)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    on_gpu = dist.get_backend(group) != "gloo"
    b = _exchange_buffers(ctx, p.capacity, world, on_gpu)
    ts = ctx.torch_stream
    if on_gpu and ts is not None:
        with torch.cuda.stream(ts):
            dist.all_gather_into_tensor(b["gathered"], b["send"], group=group)
            b["host"].copy_(b["gathered"], non_blocking=True)
            b["event"].record(ts)
        while not b["event"].query():
            pass
    elif on_gpu:
        ctx.synchronize()
        dist.all_gather_into_tensor(b["gathered"], b["send"], group=group)
        b["host"].copy_(b["gathered"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
    else:
        ctx.synchronize()
        dist.all_gather_into_tensor(b["gathered"], b["send"].cpu(), group=group)


def _sharded_over_bounds(ctx: Context, p: SearchParams, local_to_global: Sequence[int], num_files_total: int,
                         group=None) -> list[N.Match] | None:
    """numPerFile = 1 (see _search_over_bounds): every rank finds the exact maxima of its own files, the
    ranks exchange them (one all-gather of the first 1023 records + a header holding the count; a second,
    full one only if some rank has more) and replay the selection over all files in global order."""
    import torch
    import torch.distributed as dist

    nloc = ctx.num_files
    cap = p.capacity
    world = dist.get_world_size(group)
    on_gpu = dist.get_backend(group) != "gloo"
    maps = _rank_maps(ctx, local_to_global, group)
    b = _exchange_buffers(ctx, cap, world, on_gpu)
    rec, rows = b["rec"], b["rows"]
    ts = ctx.torch_stream
    # First with every rank leaving out the files that cannot reach the numMatches-th best of its own files'
    # bounds (corr_compact_top_device; the bound travels in the header row); where the finished selection shows
    # that this was not sound (every rank sees the same gathered data and decides the same), with every file.
    for top in (True, False):
        if nloc:
            if top:
                # (nothing of the run is read but its candidates: the spectral kernel then lists them itself)
                ctx.set_screen_output(True)
                try:
                    ctx.corr_run_screen_f16(0, nloc)
                finally:
                    ctx.set_screen_output(False)
                # row 0 = header (count and bound, written by the library), rows 1.. = records
                ctx.corr_compact_top_device(0, nloc, p.num_matches, b["rec_ptr"], cap, b["count_ptr"])
            else:
                ctx.corr_compact_bounded_device(0, nloc, nloc, b["rec_ptr"], cap, b["count_ptr"])
            ctx.corr_rescore_device(b["rec_ptr"], cap, b["count_ptr"])
        else:
            # fewer files than ranks: this rank has none, but takes part in the exchange with a count of zero
            # (and a bound of -inf: it leaves nothing out)
            rec[0].zero_()
            rec[0, 2] = -8388608  # the bits of float -inf
            torch.cuda.current_stream().synchronize()
        if on_gpu and ts is not None:
            # the context runs on a stream torch owns (Context.on_torch_stream): collective and copy are queued
            # behind the kernels on the device, one host-side wait at the end
            with torch.cuda.stream(ts):
                dist.all_gather_into_tensor(b["gathered"], b["send"], group=group)
                b["host"].copy_(b["gathered"], non_blocking=True)
                b["event"].record(ts)
            while not b["event"].query():  # polling: a blocking wait costs its wake-up on top of a 0.3 ms search
                pass
        elif on_gpu:
            # the context owns its stream: a host-side wait, then the collective on torch's stream.  (Wrapping the
            # library's stream in torch.cuda.ExternalStream leaves torch and RCCL holding a stream the context
            # destroys: the process then crashes on exit.)
            ctx.synchronize()
            dist.all_gather_into_tensor(b["gathered"], b["send"], group=group)
            b["host"].copy_(b["gathered"], non_blocking=True)
            torch.cuda.current_stream().synchronize()
        else:
            ctx.synchronize()
            dist.all_gather_into_tensor(b["gathered"], b["send"].cpu(), group=group)
        h = b["view"]
        sel = Selector(p.num_matches, p.num_per_file, p.min_spacing, p.step_size, ctx.target_frames)
        try:
            # translation to global file ids, merge, sort and replay in one native call
            if top:
                ok = sel.feed_gathered_top(h, world, rows, maps["flat"], maps["offsets"])
            else:
                ok = sel.feed_gathered(h, world, rows, maps["flat"], maps["offsets"], num_files_total, num_files_total)
            if ok is None:  # rare: some rank has more records than the first exchange carries
                counts = h[:, 0, :2].copy().view(np.int64).reshape(world)
                if (counts > cap).any():
                    return None  # every rank sees the overflow
                full = [torch.empty_like(rec) for _ in range(world)] if on_gpu else [
                    torch.empty((cap + 1, 4), dtype=torch.int32) for _ in range(world)]
                if ts is not None:
                    ts.synchronize()
                dist.all_gather(full, rec if on_gpu else rec.cpu(), group=group)
                out = []
                for r in range(world):
                    part = full[r][1: 1 + int(counts[r])].cpu().numpy()
                    c = np.ascontiguousarray(part, dtype=np.int32).view(N.CANDIDATE_DTYPE).reshape(-1).copy()
                    c["file"] = maps["per_rank"][r][c["file"]]
                    out.append(c)
                cands = _sort_candidates(np.concatenate(out) if out else np.zeros(0, N.CANDIDATE_DTYPE))
                if top:
                    ok = sel.feed_top(cands, float(h[:, 0, 2].copy().view(np.float32).max()))
                else:
                    ok = sel.feed_screened(cands, num_files_total, num_files_total)
            if ok:
                return sel.result()
        finally:
            sel.close()
    return None


def _sharded_exact_maxima(ctx: Context, p: SearchParams, local_to_global: Sequence[int], num_files_total: int,
                           group=None) -> list[N.Match]:
    """numPerFile = 1 without the half-precision kernel's help: every rank runs the exact kernel over its own
    files, takes the first-occurrence maximum of each (dense read-back) and the ranks exchange those."""
    import torch

    nloc = ctx.num_files
    ctx.corr_run(0, nloc)
    dense = ctx.corr_read_dense_range(0, nloc) if nloc else []
    parts = []
    for f, (sim, boost) in enumerate(dense):
        if len(sim) == 0:
            continue
        nan = np.isnan(sim)
        if nan[0]:
            # A NaN similarity (a window without spread) can only ever be offered as the first window of its
            # file, while the sets still have room (entryHasSpace, :124-128; `sim > lowestSim` is false for
            # NaN, :234), and then blocks everything after it: such a file travels densely, whose replay is
            # exact whatever the state of the sets.  Nothing here is decided per rank: a list that grows
            # beyond the capacity is seen by every rank in the exchange.
            part = np.zeros(len(sim), dtype=N.CANDIDATE_DTYPE)
            part["offset"] = np.arange(len(sim), dtype=np.int32)
            part["sim"], part["boost"] = sim, boost
        else:
            # later NaN windows are never offered: the file contributes its largest similarity, first occurrence
            o = int(np.nanargmax(sim)) if nan.any() else int(np.argmax(sim))
            part = np.zeros(1, dtype=N.CANDIDATE_DTYPE)
            part["offset"], part["sim"], part["boost"] = o, sim[o], boost[o]
        part["file"] = local_to_global[f]
        parts.append(part)
    rows = np.concatenate(parts) if parts else np.zeros(0, dtype=N.CANDIDATE_DTYPE)
    rec_h = np.zeros(max(p.capacity, 1), dtype=N.CANDIDATE_DTYPE)
    rec_h[: min(len(rows), len(rec_h))] = rows[: len(rec_h)]
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    rec = torch.from_numpy(rec_h.view(np.int32).reshape(-1, 4).copy()).to(dev)
    cnt = torch.tensor([len(rows)], dtype=torch.int64, device=dev)
    cands = exchange_candidates(rec, cnt, p.capacity, group)
    if cands is None:
        raise RuntimeError(f"candidate buffer overflow (> {p.capacity}); raise SearchParams.capacity")
    sel = Selector(p.num_matches, p.num_per_file, p.min_spacing, p.step_size, ctx.target_frames)
    try:
        # files without windows contribute nothing: feed file by file instead of sgz_sel_feed_screened
        for g, part in iter_files(cands):
            sel.feed_candidates(g, part["offset"], part["sim"], part["boost"])
        return sel.result()
    finally:
        sel.close()


def search_sharded(ctx: Context, p: SearchParams, head_global: Sequence[int], tail_global: Sequence[int],
                   group=None, num_files_total: int | None = None) -> list[N.Match]:
    """Multi-process search, one process per GPU.  The context holds the head
    files first (global ids `head_global`, identical on every rank) and then this
    rank's tail files (`tail_global`, ascending).  `num_files_total` is the number
    of files over all ranks (global ids 0 .. num_files_total-1); it enables the
    search over bounds for numPerFile = 1."""
    nh, nt = len(head_global), len(tail_global)
    assert ctx.num_files == nh + nt
    # local index -> global id, built once per placement: the two sequences describe the files the context holds, and
    # a new placement means a new database (Context.db_reset drops this), so the same two objects with the same
    # lengths are the same placement -- a search must not pay for thousands of list entries each time
    memo = ctx.__dict__.get("_placement")
    if memo is None or memo[0] is not head_global or memo[1] is not tail_global or memo[2] != (nh, nt):
        memo = (head_global, tail_global, (nh, nt), tuple(head_global) + tuple(tail_global))
        ctx.__dict__["_placement"] = memo
    l2g = memo[3]
    if num_files_total is not None and num_files_total > 0:
        _exchange_ring_images(ctx, l2g, group)
    mode = _screen_mode(ctx, p) if p.screen else None
    if (p.screen and p.screen_head and p.num_per_file == 1 and p.num_matches >= 1 and mode == "f16"
            and num_files_total is not None and num_files_total > 0):  # (nothing rank-dependent: a collective follows)
        out = _sharded_over_bounds(ctx, p, l2g, num_files_total, group)
        if out is not None:  # every rank holds the same candidates, so all ranks agree on this
            return out
    if nh == 0 and num_files_total is not None and num_files_total > 0:
        # no replicated head files to derive a common threshold from
        if p.num_per_file != 1:
            raise RuntimeError("a sharded search with numPerFile > 1 needs replicated head files")
        return _sharded_exact_maxima(ctx, p, list(tail_global), num_files_total, group)
    sel = Selector(p.num_matches, p.num_per_file, p.min_spacing, p.step_size, ctx.target_frames)
    ctx.corr_run(0, nh)
    margin = p.margin if p.margin is not None or not p.screen else MARGIN[mode]
    if nt:
        if p.screen:
            _run_screen(ctx, mode, nh, nh + nt)
        else:
            ctx.corr_run(nh, nh + nt)
    _feed_dense_head(ctx, sel, range(nh), head_global)
    if not sel.is_full:  # every rank replayed the same head, so all ranks agree on this
        raise RuntimeError("head files did not fill the match set; enlarge the head (degenerate equal sims)")
    cap = p.capacity
    rec, cnt, dev = _device_buffers(cap)
    thr = sel.threshold - margin if p.screen else sel.threshold
    ctx.corr_compact_device(nh, nh + nt, thr, rec.data_ptr(), cap, cnt.data_ptr())
    if p.screen:
        ctx.corr_rescore_device(rec.data_ptr(), cap, cnt.data_ptr())
    ctx.synchronize()
    _to_global_ids(rec, cnt, cap, l2g, dev)
    cands = exchange_candidates(rec, cnt, cap, group)
    if cands is None:
        raise RuntimeError(f"candidate buffer overflow (> {cap}); raise SearchParams.capacity")
    out = finish_selection(sel, cands)
    sel.close()
    return out
