#!/bin/bash
# Static look at what hipcc made of the kernels (no GPU needed): per kernel LDS bytes, scratch, VGPRs, spills, and -- for the kernels whose
# names match $2 -- the opcode mix of the gfx950 code.  usage: tools/isa_stats.sh [build/obj | build/variants/<name>/obj] [name-regex]
cd "$(dirname "$0")/.." || exit 1
obj=${1:-build/obj}; pat=${2:-}
LLVM=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d); trap 'rm -rf "$tmp"' EXIT
for o in "$obj"/*.hip.o; do
  b=$(basename "$o" .hip.o); cp "$o" "$tmp/$b.o"
  (cd "$tmp" && $LLVM/llvm-objdump --offloading "$b.o" > /dev/null 2>&1)
  co=$(ls "$tmp" | grep "^$b.o.*gfx950" | head -1); [ -n "$co" ] || continue
  $LLVM/llvm-readelf --notes "$tmp/$co" | awk -v F="$b" '/group_segment_fixed_size/{g=$2} /\.name:/{n=$2} /private_segment_fixed_size/{p=$2} /\.sgpr_count/{s=$2} /\.vgpr_count/{v=$2} /vgpr_spill_count/{printf "%-10s lds %6d scratch %4d sgpr %3d vgpr %3d spill %2d  %s\n", F, g, p, s, v, $2, n}' | c++filt | sed 's/(.*//' 
  if [ -n "$pat" ]; then
    $LLVM/llvm-objdump -d "$tmp/$co" | c++filt > "$tmp/$b.s"
    awk -v P="$pat" '/^[0-9a-f]+ </{f = ($0 ~ P); if (f) print "== " $2} f && !/^[0-9a-f]+ </' "$tmp/$b.s" > "$tmp/$b.sel"
    if [ -s "$tmp/$b.sel" ]; then
      grep "^== " "$tmp/$b.sel"
      grep -v "^== " "$tmp/$b.sel" | grep -oE "^\s+[a-z_0-9]+" | sed 's/\s//g' | sort | uniq -c | sort -rn | head -30
    fi
  fi
done
