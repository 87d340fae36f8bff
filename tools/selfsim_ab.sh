#!/bin/bash
# Kernel times of the self-similarity of a 30-minute file with the built library and with every library in tools/variants/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for f in strugatzki_amd/lib/libstrugatzki_amd.so tools/variants/lib_*.so; do
  [ -f $f ] || continue
  echo "== $f"
  rm -rf gpurun_out/prof_ssab
  SGZ_LIB_PATH=$GRAFT_REPO_ROOT/$f timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ssab -o ss -- python3 tools/selfsim_timing.py > gpurun_out/ssab.log 2>&1 || exit 1
  python3 tools/rocpd_kernel_stats.py gpurun_out/prof_ssab/ss_results.db | grep -E "selfsim|kernel,calls" | cut -c1-110
done
rm -rf gpurun_out/prof_ssab
