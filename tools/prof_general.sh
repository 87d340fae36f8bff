#!/bin/bash
# The general search path (numPerFile > 1) under the kernel trace.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gen
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gen -o gen -- python3 tools/general_path_timing.py > gpurun_out/prof_gen.log 2>&1
grep numPerFile gpurun_out/prof_gen.log
