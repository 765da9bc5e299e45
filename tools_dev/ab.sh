#!/bin/bash
# A/B of two builds of the library on ONE box (box-to-box spread is +-3 %): tools_dev/ab/libscpose_base.so vs the in-tree build.
# usage: tools_dev/ab.sh [rounds]   -> forward W48 384^2 batch 256 and the three MFMA-bound 3x3 layer classes, alternating
cd "$(dirname "$0")/.."
R=${1:-2}
for i in $(seq $R); do
  for lib in tools_dev/ab/libscpose_base.so spacecraft-pose-estimation_amd/libscpose_hip.so; do
    echo "== $lib"
    SCPOSE_DEV=1 SCPOSE_LIB=$lib python tools_dev/time_forward.py w48 256 | tail -1
    for shape in "96 96 3 1 48 256 res" "192 192 3 1 24 256 res" "384 384 3 1 12 256 res"; do
      SCPOSE_DEV=1 SCPOSE_LIB=$lib python tools_dev/time_conv.py $shape | tail -1
    done
  done
done
