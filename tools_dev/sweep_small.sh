#!/bin/bash
N=${1:-64}
cd $GRAFT_REPO_ROOT
for shape in "128 128 3 1 16" "256 256 3 1 8" "128 128 3 1 24" "256 256 3 1 12" "192 192 3 1 24" "384 384 3 1 12" "96 96 3 1 48"; do
  SCPOSE_DEV=1 SCPOSE_DBG=32 python3 tools_dev/time_conv.py $shape $N 2>&1 | tail -2
done
python3 tools_dev/time_graph.py w32 64
python3 tools_dev/time_graph.py w32 1
python3 tools_dev/time_graph.py w48 64 384
python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_hrnet.py -m gpu -x -q 2>&1 | tail -3
