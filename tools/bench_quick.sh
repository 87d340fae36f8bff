#!/bin/bash
# Two short bench runs (default workload, one rank's shard of the large configuration) for before/after comparisons:
# ms per query (target on the GPU), kernel ms by events, ms per search alone, ms per query with the target in host memory.
cd $GRAFT_REPO_ROOT
show() { python3 -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['ms_per_step'], r['kernel_ms'], r['same_target']['ms_per_step'], r.get('host_target', {}).get('ms_per_step'), r['frac'], d['verified'], d['targets_verified'])" $1; }
timeout -k 10 200 python3 bench.py --steps 4000 --warmup 200 --cpu-seconds 0 --exact-steps 0 --other-steps 0 --steady-steps 0 > gpurun_out/q3.json 2>gpurun_out/q3.err && show gpurun_out/q3.json && \
timeout -k 10 200 python3 bench.py --target-seconds 60 --files-per-gpu 1500 --steps 400 --warmup 40 --cpu-seconds 0 --exact-steps 0 --other-steps 0 --steady-steps 0 --same-target-steps 400 > gpurun_out/q5.json 2>gpurun_out/q5.err && show gpurun_out/q5.json
