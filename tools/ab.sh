#!/bin/bash
# Same-box A/B of library variants (development aid): bash tools/ab.sh "<python args...>" var1 var2 ... ; variants are _ab/<var>.so
# Alternates the variants ROUNDS times so that clock drift of the box shows up as spread, not as a difference.
CMD="$1"; shift
LIB=halo-accumulation_amd/libhalo_hip.so
cp $LIB /tmp/keep.so
for r in 1 2 3; do
  for v in "$@"; do
    cp _ab/$v.so $LIB
    echo -n "$v: "; timeout -k 10 200 python $CMD 2>/dev/null | tail -n 1
  done
done
cp /tmp/keep.so $LIB
