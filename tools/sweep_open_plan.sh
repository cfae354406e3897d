#!/bin/bash
# open + check at n = 2^20 under different window plans for the IPA's half-zero L/R MSMs (HALO_PLAN = "lg:c,..."; development aid)
for plan in "" "18:13" "18:12" "16:11" "16:10" "14:8" "14:9" "14:7" "18:13,16:11,14:8" "18:13,16:10,14:8" "18:12,16:10,14:7"; do
  echo "PLAN=$plan $(HALO_PLAN=$plan python tools/open_loop.py 20 6 2>/dev/null | tail -3 | awk '{print $2}' | tr '\n' ' ')"
done
