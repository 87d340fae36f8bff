cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3 -o c3 -- python3 bench.py --steps 300 --warmup 20 --cpu-seconds 0 --exact-steps 0 > gpurun_out/prof_c3.json 2> gpurun_out/prof_c3.err && \
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -o c5 -- python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0 --exact-steps 0 --target-seconds 60 --files-per-gpu 1500 > gpurun_out/prof_c5.json 2> gpurun_out/prof_c5.err
