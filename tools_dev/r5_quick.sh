#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r5_quick}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_validate_golden.py tests/test_gpu_cms.py -m gpu -x -q -s -k "validate or module_forward_decode" 2>&1 | tail -12 | tee $out/tests.txt
