#!/bin/bash
# What the band-energy walk and the DCT cost in the fftSize-1024 extraction kernel: timing builds with one or both
# left out (-DSGZ_EX_PROBE=1 / 2 / 3; 0 = the kernel as it is), linked against the objects of the regular build.
# Build here (no GPU needed):   tools/extract_phase_probe.sh build
# Run on the GPU box:           tools/extract_phase_probe.sh run
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/variants
if [ "$1" = build ]; then
  mkdir -p $OUT
  make -C $ROOT/strugatzki_amd/csrc -j8 > /dev/null
  cd $ROOT/strugatzki_amd/csrc
  for v in ${PROBES:-0 1 2 3}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$ROOT/include \
      -DSGZ_EX_PROBE=$v -c sgz_extract.hip -o /tmp/sgz_exprobe_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_exprobe$v.so \
      $(ls $ROOT/build/obj/*.o | grep -v sgz_extract) /tmp/sgz_exprobe_$v.o -lz -pthread -Wl,--exclude-libs,ALL
  done
else
  cd $ROOT
  for v in ${PROBES:-0 1 2 3}; do
    echo "probe $v"
    SGZ_LIB_PATH=$OUT/lib_exprobe$v.so timeout -k 10 100 python3 tools/extract_timing.py 52920000 | grep -v int16 | tail -1 || exit 1
  done
fi
