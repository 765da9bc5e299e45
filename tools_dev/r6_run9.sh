#!/bin/bash
# round 6, GPU run 9: worker start-up with the cached record-list pickle (same harness as run 8), loader batch 16
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_run9}; mkdir -p $out
python bench.py --pipeline --pipeline-quick --pipeline-frames 32768 --pipeline-batch 16 > $out/pipeline_32768_b16.json 2> $out/pipeline.err
python -c "
import json; d=json.load(open('$out/pipeline_32768_b16.json')); print('loader batch', d['batch'], 'loader_fps', d['loader_fps'], d['worker_startup_plus_first_batch_s']); [print('  ', k, v) for k, v in d['pipeline'].items()]"
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -2
