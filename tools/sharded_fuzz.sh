#!/bin/bash
# Random sharded searches: WORLD ranks (processes sharing the one GPU, gloo exchange) against the oracle.
# usage: tools/sharded_fuzz.sh WORLD CASES SEED
set -e
W=${1:-2}; C=${2:-40}; S=${3:-1}; PORT=$((29600 + RANDOM % 300))
pids=()
for ((r = 0; r < W; r++)); do
    timeout -k 10 300 python3 tests/sharded_rank.py $PORT fuzz $r $W gloo $C $S > gpurun_out/sharded_fuzz_$r.log 2>&1 &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
tail -n 3 gpurun_out/sharded_fuzz_*.log
exit $rc
