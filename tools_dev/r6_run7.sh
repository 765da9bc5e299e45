#!/bin/bash
# round 6, GPU run 7: files -> pred.mat at loader batch 16 against 256 on a scene long enough (32768 annotations over 64 JPEG files) for the
# start-up-inclusive rate to be comparable across loader batch sizes ("within 5 % of the batch-256 figure")
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_run7}; mkdir -p $out
for b in 16 256; do
  python bench.py --pipeline --pipeline-frames 32768 --pipeline-batch $b > $out/pipeline_32768_b$b.json 2> $out/pipeline.err
  python -c "
import json; d=json.load(open('$out/pipeline_32768_b$b.json')); print('loader batch', d['batch'], 'loader_fps', d['loader_fps']); [print('  ', k, v) for k, v in d['pipeline'].items()]"
done
