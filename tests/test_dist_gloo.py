"""world_size-2 gloo test of the sharded search (runs on CPU).

The GPU part of a rank (dense similarities, compaction) is stood in for by the
oracle's dense arrays; what is exercised is the product's distributed logic:
file sharding, head replay, candidate exchange (the one all-gather) and the
merged tail replay -- which must reproduce the single-process result exactly.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import STEP, O, corr_config, make_plants, norms_default, synth_db, synth_target


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from strugatzki_amd import _native as N
        from strugatzki_amd.engine import Selector, match_tuple
        from strugatzki_amd.search import exchange_candidates, finish_selection, shard_files

        num_matches, num_per_file, min_spacing, nf, L, sims, boosts = case
        head = num_matches
        sel = Selector(num_matches, num_per_file, min_spacing, STEP, L)
        for g in range(head):  # every rank replays all replicated head files densely
            sel.feed_dense(g, sims[g], boosts[g])
        assert sel.is_full
        thr = np.float32(sel.threshold)
        cap = 4096
        rec = np.zeros(cap, dtype=N.CANDIDATE_DTYPE)
        n = 0
        for g in shard_files(nf, head, rank, world):  # this rank's slice of the tail
            idx = np.flatnonzero(sims[g] > thr)
            rec["file"][n:n + len(idx)] = g
            rec["offset"][n:n + len(idx)] = idx
            rec["sim"][n:n + len(idx)] = sims[g][idx]
            rec["boost"][n:n + len(idx)] = boosts[g][idx]
            n += len(idx)
        rec_t = torch.from_numpy(rec.view(np.int32).reshape(cap, 4).copy())
        cnt_t = torch.tensor([n], dtype=torch.int64)
        cands = exchange_candidates(rec_t, cnt_t, cap)
        got = finish_selection(sel, cands)
        ret[rank] = [match_tuple(m) for m in got]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_matches,num_per_file,min_spacing", [(6, 1, 0), (5, 2, 30 * STEP)])
def test_two_rank_exchange_reproduces_single_process_result(num_matches, num_per_file, min_spacing):
    L, ts, nf, frames = 64, 7, 20, 500
    plants = make_plants(6, L, nf, frames, ts)
    db = synth_db(nf, frames, plants)
    tgt = synth_target(ts, L)
    cfg = corr_config(ts, L, num_matches=num_matches, num_per_file=num_per_file, min_spacing=min_spacing)
    want, sims, boosts = O.correlation_run(cfg, tgt, norms_default(), db)
    case = (num_matches, num_per_file, min_spacing, nf, L, sims, boosts)
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), case, ret), nprocs=world, join=True)
    expect = [O.match_tuple(w) for w in want]
    assert ret[0] == expect and ret[1] == expect


def _worker_bounds(rank, world, port, case, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from strugatzki_amd import _native as N
        from strugatzki_amd.engine import Selector, match_tuple
        from strugatzki_amd.search import exchange_candidates, shard_files

        num_matches, nf, L, sims, boosts, cap, overflow_rank = case
        head = num_matches
        rec = np.zeros(cap, dtype=N.CANDIDATE_DTYPE)
        n = 0
        # the replicated head files and this rank's slice of the tail: per file, the windows a screening
        # bound of 2e-3 could not tell from the maximum
        for g in list(range(head)) + shard_files(nf, head, rank, world):
            idx = np.flatnonzero(sims[g] >= sims[g].max() - np.float32(2e-3))
            rec["file"][n:n + len(idx)] = g
            rec["offset"][n:n + len(idx)] = idx
            rec["sim"][n:n + len(idx)] = sims[g][idx]
            rec["boost"][n:n + len(idx)] = boosts[g][idx]
            n += len(idx)
        rec_t = torch.from_numpy(rec.view(np.int32).reshape(cap, 4).copy())
        cnt_t = torch.tensor([cap + 5 if rank == overflow_rank else n], dtype=torch.int64)
        cands = exchange_candidates(rec_t, cnt_t, cap)
        if cands is None:
            ret[rank] = None
            return
        assert len(np.unique(cands[["file", "offset"]])) == len(cands)  # replicated head records arrive once
        sel = Selector(num_matches, 1, 0, STEP, L)
        assert sel.feed_screened(cands, nf, nf)
        ret[rank] = [match_tuple(m) for m in sel.result()]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overflow_rank", [-1, 1])
def test_two_rank_search_over_bounds(overflow_rank):
    """numPerFile = 1: the ranks exchange the exact per-file maxima (plus near-maxima) and replay all files."""
    L, ts, nf, frames, num_matches = 64, 7, 20, 500, 6
    db = synth_db(nf, frames, make_plants(6, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    cfg = corr_config(ts, L, num_matches=num_matches, num_per_file=1)
    want, sims, boosts = O.correlation_run(cfg, tgt, norms_default(), db)
    case = (num_matches, nf, L, sims, boosts, 4096, overflow_rank)
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_bounds, args=(world, _free_port(), case, ret), nprocs=world, join=True)
    if overflow_rank >= 0:  # every rank learns of the overflow and takes the other path
        assert ret[0] is None and ret[1] is None
    else:
        expect = [O.match_tuple(w) for w in want]
        assert ret[0] == expect and ret[1] == expect


def test_feed_screened_matches_dense_feeding():
    """Feeding only the per-file maximum equals feeding every window, for numPerFile = 1 (with and without
    minSpacing), and the shortcut refuses what it cannot decide."""
    from strugatzki_amd import _native as N
    from strugatzki_amd.engine import Selector, match_tuple
    L, ts, nf, frames = 48, 5, 9, 300
    db = synth_db(nf, frames, make_plants(3, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    for num_matches, min_spacing in [(4, 0), (4, 50 * STEP), (20, 0), (1, 0)]:
        cfg = corr_config(ts, L, num_matches=num_matches, num_per_file=1, min_spacing=min_spacing)
        want, sims, boosts = O.correlation_run(cfg, tgt, norms_default(), db)
        rows = []
        for g in range(nf):
            for o in np.flatnonzero(sims[g] >= sims[g].max() - np.float32(1e-3)):
                rows.append((g, o, sims[g][o], boosts[g][o]))
        cands = np.array(rows, dtype=N.CANDIDATE_DTYPE)
        sel = Selector(num_matches, 1, min_spacing, STEP, L)
        assert sel.feed_screened(cands, nf, nf)
        assert [match_tuple(m) for m in sel.result()] == [O.match_tuple(w) for w in want]
        sel.close()
    # a file without candidates, and numPerFile != 1, are refused
    sel = Selector(4, 1, 0, STEP, L)
    assert not sel.feed_screened(cands[cands["file"] != 3], nf, nf)
    sel.close()
    sel = Selector(4, 2, 0, STEP, L)
    assert not sel.feed_screened(cands, nf, nf)
    sel.close()


def test_feed_gathered_equals_feed_screened_on_the_merged_list():
    """Blocks of three ranks (files dealt round robin, rank-local file indices, unsorted records) give the
    list the merged, sorted candidates give; a rank with more records than a block holds is reported."""
    from strugatzki_amd import _native as N
    from strugatzki_amd.engine import Selector, match_tuple
    L, ts, nf, frames, world = 48, 5, 10, 300, 3
    db = synth_db(nf, frames, make_plants(3, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    cfg = corr_config(ts, L, num_matches=4, num_per_file=1)
    want, sims, boosts = O.correlation_run(cfg, tgt, norms_default(), db)
    rows = 64
    blocks = np.zeros((world, rows, 4), dtype=np.int32)
    per_rank = [[g for g in range(nf) if g % world == r] for r in range(world)]
    rng = np.random.default_rng(3)
    for r in range(world):
        rec = []
        for loc, g in enumerate(per_rank[r]):
            for o in np.flatnonzero(sims[g] >= sims[g].max() - np.float32(1e-3)):
                rec.append((loc, o, sims[g][o], boosts[g][o]))
        rec = np.array(rec, dtype=N.CANDIDATE_DTYPE)[rng.permutation(len(rec))]
        assert len(rec) < rows
        blocks[r, 0, :2] = np.array([len(rec)], dtype=np.int64).view(np.int32)
        blocks[r, 1:1 + len(rec)] = rec.view(np.int32).reshape(-1, 4)
    flat = np.array(sum(per_rank, []), dtype=np.int32)
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in per_rank])]).astype(np.int64)
    sel = Selector(4, 1, 0, STEP, L)
    assert sel.feed_gathered(blocks, world, rows, flat, offsets, nf, nf) is True
    assert [match_tuple(m) for m in sel.result()] == [O.match_tuple(w) for w in want]
    sel.close()
    # replicated head files: ranks 1 and 2 also hold (and report) rank 0's first file, as their last local file
    dup = blocks.copy()
    per_rank2 = [list(x) for x in per_rank]
    g0 = per_rank[0][0]
    rec0 = np.array([(0, o, sims[g0][o], boosts[g0][o]) for o in np.flatnonzero(sims[g0] >= sims[g0].max() - np.float32(1e-3))],
                    dtype=N.CANDIDATE_DTYPE)
    for r in (1, 2):
        n_r = int(dup[r, 0, :2].copy().view(np.int64)[0])
        extra = rec0.copy()
        extra["file"] = len(per_rank2[r])
        per_rank2[r].append(g0)
        dup[r, 1 + n_r:1 + n_r + len(extra)] = extra.view(np.int32).reshape(-1, 4)
        dup[r, 0, :2] = np.array([n_r + len(extra)], dtype=np.int64).view(np.int32)
    flat2 = np.array(sum(per_rank2, []), dtype=np.int32)
    offsets2 = np.concatenate([[0], np.cumsum([len(x) for x in per_rank2])]).astype(np.int64)
    sel = Selector(4, 1, 0, STEP, L)
    assert sel.feed_gathered(dup, world, rows, flat2, offsets2, nf, nf) is True
    assert [match_tuple(m) for m in sel.result()] == [O.match_tuple(w) for w in want]
    sel.close()
    blocks[1, 0, :2] = np.array([rows], dtype=np.int64).view(np.int32)  # more than rows - 1
    sel = Selector(4, 1, 0, STEP, L)
    assert sel.feed_gathered(blocks, world, rows, flat, offsets, nf, nf) is None
    sel.close()


def _top_pruned_candidates(sims, boosts, files, num_top, err):
    """What sgz_corr_compact_top leaves of `files` (global ids): bounds = exact +- err; T = num_top-th largest lower
    bound of these files; windows with upper bound >= max(file's largest lower bound, T)."""
    from strugatzki_amd import _native as N
    lows = sorted((np.float32(sims[g].max() - err) for g in files), reverse=True)
    T = np.float32(lows[num_top - 1]) if len(lows) >= num_top else np.float32(-np.inf)
    rows = []
    for loc, g in enumerate(files):
        gate = max(np.float32(sims[g].max() - err), T)
        for o in np.flatnonzero(sims[g] + np.float32(err) >= gate):
            rows.append((loc, o, sims[g][o], boosts[g][o]))
    return np.array(rows, dtype=N.CANDIDATE_DTYPE), float(T)


@pytest.mark.parametrize("seed", range(6))
def test_feed_top_equals_the_sequential_loop_or_declines(seed):
    """Leaving out the files that cannot reach the numMatches-th best per-file bound (sgz_corr_compact_top /
    sgz_sel_feed_top): the same matches as the reference loop over every window, also with fewer files than
    matches; with duplicated files (equal similarities collapse into one set entry and leave room for a lesser
    file) the call must decline instead of returning a shorter or different list."""
    from strugatzki_amd.engine import Selector, match_tuple
    rng = np.random.default_rng(100 + seed)
    L, ts, frames = 48, 5, 300
    nf = int(rng.integers(3, 14))
    num_matches = int(rng.integers(1, 8))
    db = synth_db(nf, frames, make_plants(min(3, nf), L, nf, frames, ts))
    dup = seed % 3 == 2 and nf >= 3
    if dup:  # the two best files become identical (the plants sit in the first files)
        db[1] = db[0].copy()
    tgt = synth_target(ts, L)
    cfg = corr_config(ts, L, num_matches=num_matches, num_per_file=1)
    want, sims, boosts = O.correlation_run(cfg, tgt, norms_default(), db)
    cands, T = _top_pruned_candidates(sims, boosts, list(range(nf)), num_matches, 1e-3)
    sel = Selector(num_matches, 1, 0, STEP, L)
    ok = sel.feed_top(cands, T)
    if ok:
        assert [match_tuple(m) for m in sel.result()] == [O.match_tuple(w) for w in want]
    else:
        assert dup and num_matches >= 2 and nf > num_matches
    if not dup or num_matches < 2:
        assert ok
    sel.close()


def test_feed_top_declines_when_equal_similarities_leave_room():
    """Two identical best files, numMatches = 2: the reference's set keeps one entry for both (keyed on the
    similarity) and fills the second place with a lesser file, which the filter left out."""
    from strugatzki_amd.engine import Selector, match_tuple
    L, ts, nf, frames = 48, 5, 6, 300
    db = synth_db(nf, frames, make_plants(1, L, nf, frames, ts))
    db[1] = db[0].copy()
    tgt = synth_target(ts, L)
    want, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=2, num_per_file=1), tgt, norms_default(), db)
    assert sorted(w.file for w in want)[0] == 0 and sorted(w.file for w in want)[1] > 1
    cands, T = _top_pruned_candidates(sims, boosts, list(range(nf)), 2, 1e-3)
    assert set(cands["file"]) == {0, 1}
    sel = Selector(2, 1, 0, STEP, L)
    assert sel.feed_top(cands, T) is False
    sel.close()
    cands, _ = _top_pruned_candidates(sims, boosts, list(range(nf)), nf + 1, 1e-3)  # nothing left out
    sel = Selector(2, 1, 0, STEP, L)
    assert sel.feed_top(cands, float("-inf")) is True
    assert [match_tuple(m) for m in sel.result()] == [O.match_tuple(w) for w in want]
    sel.close()


def test_feed_gathered_top_uses_the_largest_threshold():
    """Three ranks each leave out files below their own threshold (header bytes 8..11); the merged replay equals
    the reference loop, and declines when a threshold above the finished set's worst entry says that a file may
    be missing."""
    from strugatzki_amd.engine import Selector, match_tuple
    L, ts, nf, frames, world, num_matches = 48, 5, 13, 300, 3, 3
    db = synth_db(nf, frames, make_plants(3, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    want, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=num_matches, num_per_file=1), tgt,
                                           norms_default(), db)
    rows = 64
    blocks = np.zeros((world, rows, 4), dtype=np.int32)
    per_rank = [[g for g in range(nf) if g % world == r] for r in range(world)]
    for r in range(world):
        rec, T = _top_pruned_candidates(sims, boosts, per_rank[r], num_matches, 1e-3)
        assert len(rec) < rows and np.isfinite(T) and len(set(rec["file"])) < len(per_rank[r])
        blocks[r, 0, :2] = np.array([len(rec)], dtype=np.int64).view(np.int32)
        blocks[r, 0, 2] = np.array([T], dtype=np.float32).view(np.int32)[0]
        blocks[r, 1:1 + len(rec)] = rec.view(np.int32).reshape(-1, 4)
    flat = np.array(sum(per_rank, []), dtype=np.int32)
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in per_rank])]).astype(np.int64)
    sel = Selector(num_matches, 1, 0, STEP, L)
    assert sel.feed_gathered_top(blocks, world, rows, flat, offsets) is True
    assert [match_tuple(m) for m in sel.result()] == [O.match_tuple(w) for w in want]
    sel.close()
    blocks[2, 0, 2] = np.array([2.0], dtype=np.float32).view(np.int32)[0]
    sel = Selector(num_matches, 1, 0, STEP, L)
    assert sel.feed_gathered_top(blocks, world, rows, flat, offsets) is False
    sel.close()


def test_shard_files_partitions_the_tail():
    from strugatzki_amd.search import shard_files
    parts = [shard_files(37, 5, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(5, 37))
    assert max(map(len, parts)) - min(map(len, parts)) <= 1
