#!/bin/bash
# as ab_m16.sh, on post-ReLU-like input (half zeros: what the layers see inside the network)
cd "$(dirname "$0")/.."
for i in 1 2; do
  for m in 0 1; do
    for shape in "96 96 3 1 48 256 res" "96 96 3 1 48 256 x" "192 192 3 1 24 256 res" "192 192 3 1 24 256 x" "384 384 3 1 12 256 res" "384 384 3 1 12 256 x"; do
      echo -n "M16=$m  "; SCPOSE_DEV=1 SCPOSE_M16=$m ITERS=300 python3 tools_dev/time_conv.py $shape relu 2>/dev/null | tail -1
    done
  done
done
