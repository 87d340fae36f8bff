#!/bin/bash
# The round's profile set of bench.py in one go (each step its own run with its own time limit):
#   plain runs of the default workload (default flags, then the flags the round-end driver uses) and of one rank's shard of the large configuration,
#   rocprofv3 kernel statistics of both, and the four counter passes of the default workload (the profiled runs without the
#   leg that keeps several queries in flight: its launches of the screening kernel share the chip and would enter the
#   kernel's average).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
C5="--target-seconds 60 --files-per-gpu 1500"
timeout -k 10 300 python3 bench.py > gpurun_out/r_bench_line.json 2> gpurun_out/r_bench_line.err && \
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r_driver_line.json 2> gpurun_out/r_driver_line.err && \
timeout -k 10 300 python3 bench.py $C5 --steps 300 --warmup 20 --cpu-seconds 0 --exact-steps 3 --other-steps 60 > gpurun_out/r_config5_bench_line.json 2> gpurun_out/r_config5_bench_line.err && \
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3 -o c3 -- python3 bench.py --steps 300 --warmup 20 --cpu-seconds 0 --exact-steps 0 --other-steps 100 --in-flight 0 > gpurun_out/r_bench_line_profiled.json 2> gpurun_out/prof_c3.err && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -o c5 -- python3 bench.py $C5 --steps 50 --warmup 5 --cpu-seconds 0 --exact-steps 0 --other-steps 20 --in-flight 0 > gpurun_out/r_config5_bench_line_profiled.json 2> gpurun_out/prof_c5.err && \
bash tools/prof_pmc.sh c3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --exact-steps 10 --other-steps 100 --in-flight 0 && \
python3 tools/pmc_summary.py gpurun_out/r_pmc_summary.csv sq=gpurun_out/pmc_c3_sq sq2=gpurun_out/pmc_c3_sq2 rd=gpurun_out/pmc_c3_rd wr=gpurun_out/pmc_c3_wr && \
python3 tools/rocpd_kernel_stats.py gpurun_out/prof_c3/c3_results.db > gpurun_out/r_bench_kernel_stats.csv && \
python3 tools/rocpd_kernel_stats.py gpurun_out/prof_c5/c5_results.db > gpurun_out/r_config5_kernel_stats.csv && \
rm -rf gpurun_out/prof_c3 gpurun_out/prof_c5 gpurun_out/pmc_c3_sq gpurun_out/pmc_c3_sq2 gpurun_out/pmc_c3_rd gpurun_out/pmc_c3_wr
