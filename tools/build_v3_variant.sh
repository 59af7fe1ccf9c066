#!/bin/bash
# A/B builds of the one-wave-per-SIMD fused kernel (ga_forward_kernel_v3.h):  tools/build_v3_variant.sh NAME "-DFLAG=..."
# Recompiles the v3 family objects with the extra flags and links them with the regular objects -> build/variants/libacmil_v3_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); SRC=$ROOT/acmil_amd/csrc; OUT=$ROOT/build/variants; mkdir -p $OUT
name=$1; flags=$2
objs=$(ls $SRC/build/*.o | grep -v "ga_fwd3_\|/ab_")
new=""
for fam in $(sed -n 's/^GA3_FAMILY(\([0-9]*\), *\([0-9]*\), *\([0-9]*\))/\1_\2_\3/p' $SRC/ga_families3.inc); do
  IFS=_ read ND PB KP <<< "$fam"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -Wno-unused-value -DGA3_ND=$ND -DGA3_PB=$PB -DGA3_KP=$KP $flags \
      -c $SRC/ga_forward3_inst.hip -o $OUT/ga_fwd3_${fam}_$name.o &
  new="$new $OUT/ga_fwd3_${fam}_$name.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libacmil_v3_$name.so $objs $new
rm -f $new
echo "built $OUT/libacmil_v3_$name.so ($flags)"
