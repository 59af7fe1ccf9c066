#!/bin/bash
# Build timing variants of libacmil_hip.so for tools/probe_ga.py:  tools/build_variants.sh NAME:"-DFLAG ..." [NAME:FLAGS ...]
# Only the (ND=8, KP=5, split-f16) family object is recompiled with the extra flags; everything else is linked from the
# regular build (run `make -C acmil_amd/csrc` first).  Output: build/variants/libacmil_NAME.so (git-ignored, travels with gpurun).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/acmil_amd/csrc
OUT=$ROOT/build/variants
mkdir -p $OUT
FAM=${GA_VARIANT_FAMILY:-8_5_1}
IFS=_ read ND KP MODE <<< "$FAM"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  [ "$flags" == "$spec" ] && flags=""
  (
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -Wno-unused-value \
      -DGA_ND=$ND -DGA_KP=$KP -DGA_MODE=$MODE -DGA2_TOOLS -I$ROOT/tools $flags -c $SRC/ga_forward_inst.hip -o $OUT/ga_fwd_${FAM}_$name.o
  objs=$(ls $SRC/build/*.o | grep -v "ga_fwd_${FAM}.o" | grep -v "/ab_")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libacmil_$name.so $objs $OUT/ga_fwd_${FAM}_$name.o
  rm -f $OUT/ga_fwd_${FAM}_$name.o
  echo "built $OUT/libacmil_$name.so  ($flags)"
  ) &
done
wait
