#!/bin/bash
# round 6, GPU run 8: the packed loader batches (two tensors per batch instead of fifteen): the CLI / e2e / validate tests, then files -> pred.mat at
# loader batch 16 and 256 (32 workers, --pipeline-quick, 32768 annotations) against run 7's figures of the same harness
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_run8}; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_validate_golden.py tests/test_gpu_cms.py -m gpu -x -q > $out/tests.txt 2>&1; tail -3 $out/tests.txt
for b in 16 256; do
  python bench.py --pipeline --pipeline-quick --pipeline-frames 32768 --pipeline-batch $b > $out/pipeline_32768_b$b.json 2> $out/pipeline.err
  python -c "
import json; d=json.load(open('$out/pipeline_32768_b$b.json')); print('loader batch', d['batch'], 'loader_fps', d['loader_fps']); [print('  ', k, v) for k, v in d['pipeline'].items()]"
done
