#!/bin/bash
# Self-similarity of a 30-minute file under the kernel trace; tests of the path first.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
if [ "$1" = "test" ]; then
  timeout -k 10 300 python -u -m pytest tests/test_gpu_halfcorr.py -x -q --timeout 120 --timeout-method thread > gpurun_out/ss_test.log 2>&1 || { tail -30 gpurun_out/ss_test.log; exit 1; }
  tail -2 gpurun_out/ss_test.log
fi
rm -rf gpurun_out/prof_ss
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ss -o ss -- python3 tools/selfsim_timing.py > gpurun_out/prof_ss.log 2>&1
grep ext= gpurun_out/prof_ss.log
