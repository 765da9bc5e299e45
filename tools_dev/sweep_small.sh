#!/bin/bash
# Small-batch tile sweep of the 32x32x16 kernels (W32 branch shapes): tools_dev/sweep_small.sh [N]
N=${1:-64}
cd $GRAFT_REPO_ROOT
for shape in "64 64 3 1 32" "128 128 3 1 16" "256 256 3 1 8"; do
  echo "== $shape N=$N default"; SCPOSE_DEV=1 SCPOSE_DBG=32 python3 tools_dev/time_conv.py $shape $N 2>&1 | tail -2
  for occ in 1 2 3; do for nr in 1 2 3 4; do
    echo "-- occ=$occ nr=$nr"; SCPOSE_DEV=1 SCPOSE_DBG=32 SCPOSE_M32_OCC=$occ SCPOSE_M32_NR=$nr python3 tools_dev/time_conv.py $shape $N 2>&1 | tail -2
  done; done
done
