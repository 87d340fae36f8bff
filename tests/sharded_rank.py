"""One rank of the sharded search on a real GPU (started by test_gpu_correlation.py as a child process:
torch has to initialise the GPU before the library does, which a shared pytest process cannot guarantee).
Prints the matches as JSON."""
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


def main() -> int:
    port, screen_head = int(sys.argv[1]), sys.argv[2] == "1"
    L, ts, nf, frames, num_matches = 96, 17, 14, 900, 4
    db = synth_db(nf, frames, make_plants(5, L, nf, frames, ts))
    tgt = synth_target(ts, L)
    ctx = S.Context(0)
    ctx.db_reset(db[0].shape[1])
    for f in db:
        ctx.db_add_file(f)
    ctx.set_norms(norms_default())
    ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        p = S.SearchParams(num_matches, 1, 0, STEP, screen_head=screen_head)
        got = S.search_sharded(ctx, p, list(range(num_matches)), list(range(num_matches, nf)), num_files_total=nf)
    finally:
        dist.destroy_process_group()
    print(json.dumps([[m.file, m.start, m.stop, float(m.sim)] for m in got]))
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
