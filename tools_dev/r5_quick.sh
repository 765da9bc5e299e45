#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r5_quick}; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -4 | tee $out/tests.txt
timeout 1200 python bench.py --pipeline > $out/pipeline.json 2> $out/pipeline_err.txt; tail -c 2600 $out/pipeline.json
