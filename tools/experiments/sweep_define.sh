#!/bin/bash
# Kernel-tuning sweeps over compile-time constants of the NTT tiles: builds one variant of the library per argument into
# dacapo_amd/lib/variants/ (git-ignored like every .so).  Each argument is NAME:FLAGS, e.g.
#   tools/experiments/sweep_define.sh "t0l0:-DDC_TW_WORD_STAGES_T=0 -DDC_TW_WORD_STAGES_L=0" "t2l0:-DDC_TW_WORD_STAGES_T=2"
# and is timed on the GPU box with  DACAPO_AMD_LIB=dacapo_amd/lib/variants/libSEAL_HEVM.<NAME>.so python tools/experiments/ntt_leg.py
set -e
cd "$(dirname "$0")/../dacapo_amd/csrc"
mkdir -p ../lib/variants
for arg in "$@"; do
  name=${arg%%:*}; flags=${arg#*:}
  B=/tmp/sweep_build_$name
  mkdir -p $B
  for f in ntt_kernels fused_ks ntt_full; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed $flags -c $f.hip -o $B/$f.o &
  done
  wait
  objs=""
  for o in build/*.o; do
    b=$(basename $o)
    if [ -f $B/$b ]; then objs="$objs $B/$b"; else objs="$objs $o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libSEAL_HEVM.$name.so $objs -lz -ldl
  echo built $name
done
