#!/bin/bash
# Prints VGPR / spill / scratch / LDS of every kernel in libeleven_hip.so (from the code objects' metadata).
set -e
SO=${1:-$(dirname "$0")/../elevenrender_amd/libeleven_hip.so}
T=$(mktemp -d)
cp "$SO" "$T/lib.so"
(cd "$T" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null)
for f in "$T"/lib.so.*gfx950; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" 2>/dev/null | awk '
    /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {sp=$2} /\.private_segment_fixed_size:/ {pr=$2}
    /\.group_segment_fixed_size:/ {lds=$2} /\.sgpr_count:/ {sg=$2}
    /\.wavefront_size:/ {printf "%-90s vgpr %3d spill %3d scratch %5d lds %6d sgpr %3d\n", substr(name,1,90), v, sp, pr, lds, sg}'
done
rm -rf "$T"
