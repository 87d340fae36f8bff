#!/bin/bash
# What the three phases of the spectral screening kernel cost: timing builds with one phase left out each
# (-DSGZ_PROBE=1 spectra loads, 2 inverse transform, 3 epilogue; 0 = the kernel as it is), linked against the
# objects of the regular build, timed by tools/spectral_probe_timing.py at the two benchmark sizes.
# Build here (no GPU needed):   tools/spectral_phase_probe.sh build
# Run on the GPU box:           tools/spectral_phase_probe.sh run
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/variants
if [ "$1" = build ]; then
  mkdir -p $OUT
  make -C $ROOT/strugatzki_amd/csrc -j8 > /dev/null
  cd $ROOT/strugatzki_amd/csrc
  for v in ${PROBES:-0 1 2 3}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$ROOT/include \
      -DSGZ_PROBE=$v -c sgz_corr_fft.hip -o /tmp/sgz_probe_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_probe$v.so \
      $(ls $ROOT/build/obj/*.o | grep -v sgz_corr_fft) /tmp/sgz_probe_$v.o -lz -pthread -Wl,--exclude-libs,ALL
  done
else
  cd $ROOT
  for v in ${PROBES:-0 1 2 3}; do
    SGZ_LIB_PATH=$OUT/lib_probe$v.so timeout -k 10 100 python3 tools/spectral_probe_timing.py || exit 1
    SGZ_LIB_PATH=$OUT/lib_probe$v.so PROBE_L=5168 PROBE_NF=1500 timeout -k 10 100 python3 tools/spectral_probe_timing.py || exit 1
  done
fi
