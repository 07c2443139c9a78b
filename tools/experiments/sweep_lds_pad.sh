#!/bin/bash
# LDS row padding sweep for the NTT tiles (ntt_tile.hpp kLdsPad): builds one variant of the library per value into
# dacapo_amd/lib/variants/ (git-ignored like every .so), to be timed on the GPU box with
#   for p in 1 2 3 4 5 8 9; do DACAPO_AMD_LIB=dacapo_amd/lib/variants/libSEAL_HEVM.pad$p.so python tools/experiments/ntt_leg.py; done
set -e
cd "$(dirname "$0")/../dacapo_amd/csrc"
mkdir -p ../lib/variants
for p in "$@"; do
  rm -rf /tmp/sweep_build_$p && mkdir -p /tmp/sweep_build_$p
  for f in ntt_kernels fused_ks; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -DDC_LDS_PAD=$p -c $f.hip -o /tmp/sweep_build_$p/$f.o &
  done
  wait
  objs=""
  for o in build/*.o; do
    b=$(basename $o)
    if [ -f /tmp/sweep_build_$p/$b ]; then objs="$objs /tmp/sweep_build_$p/$b"; else objs="$objs $o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libSEAL_HEVM.pad$p.so $objs -lz -ldl
  echo built pad $p
done
