#!/bin/bash
# Per-kernel resource usage (VGPRs, AGPRs, scratch, occupancy, LDS) of one HIP source:  tools/kres.sh acmil_amd/csrc/linear.hip [name filter] [extra flags]
f=$1; pat=${2:-.}; shift; shift
cd "$(dirname "$f")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$(basename "$f")" -o /dev/null 2>&1 | \
  awk '/Function Name:/{n=$5} / VGPRs:/{v=$4} /AGPRs:/{a=$4} /ScratchSize/{s=$5} /Occupancy/{o=$5} /LDS Size/{print n, "vgpr", v, "agpr", a, "scratch", s, "occ", o, "lds", $6}' | c++filt | grep -E "$pat"
exit 0
