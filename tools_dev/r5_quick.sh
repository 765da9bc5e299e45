#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r5_quick}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -s -k "files_to_poses" 2>&1 | tail -8 | tee $out/e2e.txt
