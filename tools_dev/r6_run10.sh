#!/bin/bash
# round 6, GPU run 10: how many loader workers, now that a batch costs the feeding thread two descriptor hand-overs instead of fifteen? (256-core host)
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_run10}; mkdir -p $out
for w in 64 96; do
  python bench.py --pipeline --pipeline-quick --pipeline-frames 32768 --pipeline-batch 16 --pipeline-workers $w > $out/pipeline_32768_b16_w$w.json 2> $out/pipeline.err
  python -c "
import json; d=json.load(open('$out/pipeline_32768_b16_w$w.json')); print('workers', d['workers'], 'loader_fps', d['loader_fps'], d['worker_startup_plus_first_batch_s']); [print('  ', k, v) for k, v in d['pipeline'].items()]"
done
