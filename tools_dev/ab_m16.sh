#!/bin/bash
# Same-box A/B of the producer/consumer 3x3 kernel's consumer shape: SCPOSE_M16=0 (32x32x16) vs default (16x16x32), development library.
# usage: tools_dev/ab_m16.sh [rounds]
cd "$(dirname "$0")/.."
R=${1:-3}
for i in $(seq $R); do
  for m in 0 1; do
    for shape in "96 96 3 1 48 256 res" "96 96 3 1 48 256" "192 192 3 1 24 256 res" "192 192 3 1 24 256" "384 384 3 1 12 256 res" "384 384 3 1 12 256"; do
      echo -n "M16=$m  "; SCPOSE_DEV=1 SCPOSE_M16=$m ITERS=300 python3 tools_dev/time_conv.py $shape 2>/dev/null | tail -1
    done
  done
done
