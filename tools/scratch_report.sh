#!/bin/bash
# Kernels of one HIP source (at a git revision, default: the working tree) that use scratch memory or spill registers.
# usage: tools/scratch_report.sh conv_igemm.hip [rev]
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"; S="$R/unsupervised_detection_amd/csrc"; T=$(mktemp -d)
if [ -n "$2" ]; then git -C "$R" archive "$2" unsupervised_detection_amd/csrc include | tar -x -C "$T"; S="$T/unsupervised_detection_amd/csrc"; fi
X=""; [ "$1" = conv_wino.hip ] && X="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $X --cuda-device-only --no-gpu-bundle-output -c "$S/$1" -o "$T/k.co" 2> /dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$T/k.co" | grep -E "^\s+\.name:|\.private_segment_fixed_size|\.vgpr_spill_count|\.sgpr_spill_count|\.vgpr_count" | paste - - - - - |
  awk '{ n=""; for (i=1;i<=NF;i++) { if ($i==".name:") n=$(i+1); if ($i==".private_segment_fixed_size:") p=$(i+1); if ($i==".vgpr_spill_count:") v=$(i+1); if ($i==".sgpr_spill_count:") s=$(i+1); if ($i==".vgpr_count:") c=$(i+1) } if (p+v+s > 0) print "scratch", p, "vgpr_spill", v, "sgpr_spill", s, "vgprs", c, n }' | c++filt | cut -c1-200
rm -rf "$T"
