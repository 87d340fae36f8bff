"""Host-side timeline of one sharded search step on one rank (RCCL group of size 1): where the time
between the kernels and the result goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import strugatzki_amd as S
from strugatzki_amd import _native as N
import importlib
SR = importlib.import_module("strugatzki_amd.search")
from strugatzki_amd.engine import Selector
import bench as B

L, FR, NF = 2584, 25839, 120
plants = B.plant_list(10, L, [int(round(j * (NF - 1) / 9)) for j in range(10)], FR)
npl = [N.Plant(p["file"], p["offset"], p["src_file"], p["src_offset"], p["length"], p["amp"], p["gain"]) for p in plants]
ctx = S.Context.on_torch_stream(0) if os.environ.get("TS", "1") == "1" else S.Context(0)
ctx.db_reset(14)
for g in range(NF):
    ctx.db_add_synthetic(B.SEED, g, FR, npl)
ctx.set_norms(B.norms_default())
c2 = S.Context(0); c2.db_reset(14); c2.db_add_synthetic(B.SEED, B.TARGET_FILE, B.TARGET_START + L + 16, [])
tgt = c2.db_read_file(0)[B.TARGET_START:B.TARGET_START + L].copy(); c2.close()
ctx.corr_set_target(tgt, 0.5, 8.0)
p = S.SearchParams(10, 1, 0, 512)
glob = list(range(NF))
for _ in range(5):
    S.search_sharded(ctx, p, [], glob, num_files_total=NF)
cap = p.capacity
b = SR._exchange_buffers(ctx, cap, 1, True)
maps = SR._rank_maps(ctx, glob, None)
ts = ctx.torch_stream
acc = {}
def mark(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
R = 200
for _ in range(R):
    t = time.perf_counter()
    ctx.corr_run_screen_f16(0, NF); t = mark("launch run", t)
    ctx.corr_compact_bounded_device(0, NF, NF, b["rec_ptr"], cap, b["count_ptr"]); t = mark("launch compact", t)
    ctx.corr_rescore_device(b["rec_ptr"], cap, b["count_ptr"]); t = mark("launch rescore", t)
    if ts is not None:
        with torch.cuda.stream(ts):
            dist.all_gather_into_tensor(b["gathered"], b["send"]); t = mark("all_gather call", t)
            b["host"].copy_(b["gathered"], non_blocking=True); t = mark("copy call", t)
        ts.synchronize(); t = mark("sync", t)
    else:
        ctx.synchronize(); t = mark("ctx sync", t)
        dist.all_gather_into_tensor(b["gathered"], b["send"]); t = mark("all_gather call", t)
        b["host"].copy_(b["gathered"], non_blocking=True); t = mark("copy call", t)
        torch.cuda.current_stream().synchronize(); t = mark("sync", t)
    h = b["view"]
    counts = h[:, 0, :2].copy().view(np.int64).reshape(1)
    part = h[0, 1:1 + int(counts[0])]
    c = np.ascontiguousarray(part, dtype=np.int32).view(N.CANDIDATE_DTYPE).reshape(-1).copy()
    c["file"] = maps[0][c["file"]]
    cands = SR._sort_candidates(c); t = mark("numpy", t)
    sel = Selector(10, 1, 0, 512, L)
    ok = sel.feed_screened(cands, NF, NF); res = sel.result(); sel.close(); t = mark("replay", t)
tot = sum(acc.values())
print("TIMELINE", "  ".join(f"{k}={v / R * 1e6:.1f}us" for k, v in acc.items()), f" total={tot / R * 1e6:.1f}us", file=sys.stderr, flush=True)
dist.destroy_process_group()
