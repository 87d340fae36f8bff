#!/bin/bash
# Kernel time of the extraction of 20 minutes of audio with the built library and with every library in tools/variants/.
cd $GRAFT_REPO_ROOT
for f in strugatzki_amd/lib/libstrugatzki_amd.so tools/variants/lib_*.so; do
  [ -f $f ] || continue
  echo "== $f"
  SGZ_LIB_PATH=$GRAFT_REPO_ROOT/$f timeout -k 10 100 python3 tools/extract_timing.py 52920000 | grep -v int16 | tail -1 || exit 1
done
