#!/bin/bash
# Builds of the library that differ in csrc/w4a16_gemm_pk.hip only (the TCE_PK_VAR_* experiment macros), for same-box comparisons through TCE_LIB_PATH:
#   tinychatengine_amd/lib/abl/libtce_<name>.so   (git-ignored; travels with gpurun).   usage: build_pk_variants.sh name:-DFLAG[,-DFLAG...] ...
cd "$(dirname "$0")/../.." || exit 1
L=tinychatengine_amd/lib; mkdir -p $L/abl
for spec in "$@"; do
  name=${spec%%:*}; flags=$(echo "${spec#*:}" | tr ',' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -Wno-unused-function $flags -I include -I tinychatengine_amd/csrc \
      -c tinychatengine_amd/csrc/w4a16_gemm_pk.hip -o /tmp/pk_$name.o || exit 1
  objs=$(ls $L/*.o | grep -v w4a16_gemm_pk.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/libtce_$name.so $objs /tmp/pk_$name.o || exit 1
  echo built $L/abl/libtce_$name.so
done
