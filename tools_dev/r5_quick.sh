#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r5_quick}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -6 | tee $out/tests.txt
