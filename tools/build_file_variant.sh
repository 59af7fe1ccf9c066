#!/bin/bash
# Timing / A-B variant of libacmil_hip.so with ONE source file recompiled under extra flags:
#   tools/build_file_variant.sh NAME file.hip "-DFLAG=1 ..."   ->  build/variants/libacmil_NAME.so  (select with ACMIL_HIP_LIB)
# Everything else is linked from the regular build (run `make -C acmil_amd/csrc` first).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/acmil_amd/csrc
OUT=$ROOT/build/variants
mkdir -p $OUT
name=$1; file=$2; flags=$3
base=$(basename $file .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -Wno-unused-value $flags -c $SRC/$file -o $OUT/${base}_$name.o
objs=$(ls $SRC/build/*.o | grep -v "/${base}.o" | grep -v "/ab_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libacmil_$name.so $objs $OUT/${base}_$name.o
rm -f $OUT/${base}_$name.o
echo "built $OUT/libacmil_$name.so  ($file: $flags)"
