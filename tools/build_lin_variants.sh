#!/bin/bash
# Timing variants of the Linear kernels:  tools/build_lin_variants.sh NAME:"-DLIN64_ABL=1" ...  -> build/variants/libacmil_NAME.so
# (only linear.hip is recompiled; run `make -C acmil_amd/csrc` first; load with ACMIL_HIP_LIB=path)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/acmil_amd/csrc
OUT=$ROOT/build/variants
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  [ "$flags" == "$spec" ] && flags=""
  (
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -Wno-unused-value $flags -c $SRC/linear.hip -o $OUT/linear_$name.o
  objs=$(ls $SRC/build/*.o | grep -v "/linear.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libacmil_$name.so $objs $OUT/linear_$name.o
  rm -f $OUT/linear_$name.o
  echo "built $OUT/libacmil_$name.so  ($flags)"
  ) &
done
wait
