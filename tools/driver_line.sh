#!/bin/bash
# The round-end driver's command, three times: how far the 20 timed steps scatter.
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/drv_$i.json 2> gpurun_out/drv_$i.err || exit 1
python3 -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_samples'], r['frac'], r['steady']['ms_per_step'], d['verified'], d['targets_verified'])" gpurun_out/drv_$i.json
done
