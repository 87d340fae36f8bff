"""One rank of the sharded search on a real GPU (started by test_gpu_correlation.py as a child process:
torch has to initialise the GPU before the library does, which a shared pytest process cannot guarantee).
usage: sharded_rank.py PORT MODE RANK WORLD BACKEND
MODE: "bounds" (search over bounds), "general" (replicated head files, screen_head off), "exact" (no
replicated files, no screening).  Every rank prints the matches as one JSON line."""
import json
import os
import sys

import torch

torch.cuda.init()
torch.cuda.set_device(0)
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from common import STEP, make_plants, norms_default, synth_db, synth_target  # noqa: E402

import strugatzki_amd as S  # noqa: E402

CASE = dict(L=96, ts=17, nf=14, frames=900, num_matches=4, nplants=5)


def main() -> int:
    port, mode, rank, world, backend = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    q = CASE
    db = synth_db(q["nf"], q["frames"], make_plants(q["nplants"], q["L"], q["nf"], q["frames"], q["ts"]))
    tgt = synth_target(q["ts"], q["L"])
    if mode == "general":  # head files on every rank, the rest dealt round robin
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
        p = S.SearchParams(q["num_matches"], 1, 0, STEP, screen=mode != "exact", screen_head=mode == "bounds")
        got = S.search_sharded(ctx, p, head, tail, num_files_total=q["nf"])
    finally:
        dist.destroy_process_group()
    print(json.dumps([[m.file, m.start, m.stop, float(m.sim)] for m in got]))
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
