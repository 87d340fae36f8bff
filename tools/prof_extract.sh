#!/bin/bash
# Feature extraction of 1 h of mono audio under the kernel trace.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ex
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ex -o ex -- python3 tools/extract_timing.py > gpurun_out/prof_ex.log 2>&1
tail -1 gpurun_out/prof_ex.log
