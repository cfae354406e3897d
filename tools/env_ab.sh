#!/bin/bash
# Same-box comparison of environment switches (development aid): bash tools/env_ab.sh "<python args>" "VAR=val ..." "VAR2=val" ...
# ("-" = no variables); three alternating rounds, last line of each run.
CMD="$1"; shift
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    echo -n "[$v] "; env $e timeout -k 10 200 python $CMD 2>/dev/null | tail -n 1
  done
done
