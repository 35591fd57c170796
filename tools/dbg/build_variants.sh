#!/bin/bash
# Builds A/B variants of liboi_hip.so that differ in ONE translation unit's -D flags:
#   tools/dbg/build_variants.sh mlp_fwd3.hip name1 "-DX=1" name2 "-DX=2" ...
# -> object-intrinsics_amd/build/ab/liboi_<name>.so (select with OI_LIB=<path>)
set -e
R=$(cd $(dirname $0)/../.. && pwd)/object-intrinsics_amd
src=$1; shift
mkdir -p $R/build/ab
extra=""
{ [ "$src" = "mlp_fwd3.hip" ] || [ "$src" = "mlp_fwd3b.hip" ] || [ "$src" = "mlp_bwd.hip" ]; } && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
[ "$src" = "mlp_fwd3b.hip" ] && extra="$extra -fno-slp-vectorize"
others=$(ls $R/build/*.o | grep -v "/${src%.hip}.o")
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $extra $flags -c $R/csrc/$src -o $R/build/ab/${src%.hip}_$name.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/ab/liboi_$name.so $others $R/build/ab/${src%.hip}_$name.o && echo built $name ) &
done
wait
