#!/bin/bash
# Kernel-tuning sweeps over compile-time constants: builds one variant of the library per argument into dacapo_amd/lib/variants/
# (git-ignored like every .so).  Each argument is NAME:FLAGS, e.g.
#   tools/experiments/sweep_define.sh "nopm:-DDC_EXP_NO_PRIME_MAJOR" "nocr:-DDC_EXP_NO_COLS_REMAP"
# and is timed on the GPU box with  DACAPO_AMD_LIB=dacapo_amd/lib/variants/libSEAL_HEVM.<NAME>.so python tools/...
set -e
cd "$(dirname "$0")/../../dacapo_amd/csrc"
mkdir -p ../lib/variants
for arg in "$@"; do
  name=${arg%%:*}; flags=${arg#*:}
  B=/tmp/sweep_build_$name
  mkdir -p $B
  n=0
  for f in *.hip; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed $flags -c $f -o $B/${f%.hip}.o &
    n=$((n + 1)); if [ $((n % 6)) = 0 ]; then wait; fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libSEAL_HEVM.$name.so $B/*.o build/*.host.o -Wl,--version-script=exports.map -lz -ldl
  echo built $name
done
