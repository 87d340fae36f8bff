#!/bin/bash
# Transform lengths side by side (SGZ_FFT_N forces one): ms per query, kernel ms by events, ms per search alone.
# usage: fft_length_ab.sh [bench.py flags of the configuration]
cd $GRAFT_REPO_ROOT
for n in 16384 8192 4096; do
SGZ_FFT_N=$n timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --exact-steps 0 --other-steps 0 --steady-steps 0 > gpurun_out/b_$n.json 2>gpurun_out/b_$n.err || exit 1
python3 -c "
import json; d=json.loads(open('gpurun_out/b_$n.json').read().strip().splitlines()[-1]); r=d['roofline']; print($n, d['ms_per_step'], r['kernel_ms'], r['same_target']['ms_per_step'], r['transform_length'], r['blocks'], d['verified'], d['targets_verified'])"
done
