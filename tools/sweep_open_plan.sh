#!/bin/bash
# open + check at n = 2^20 under different window plans for the IPA's half-zero L/R MSMs (HALO_PLAN = "lg:c,..."; development aid)
# usage: sweep_open_plan.sh "plan1" "plan2" ...   ("" = the size-based table)
for plan in "$@"; do
  echo "PLAN=$plan $(HALO_PLAN=$plan python tools/open_loop.py 20 8 2>/dev/null | tail -5 | awk '{print $2}' | sort -n | head -3 | tr '\n' ' ')"
done
