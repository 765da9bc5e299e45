#!/bin/bash
# round 6, GPU run 3: the whole -m gpu suite on the tree with engine-batch coalescing, the chain check in the bench line, the
# cross-check tests; then the driver-style bench line and the files -> poses pipeline at the shipped YAML's batch 16
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_run3}; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; tail -5 $out/tests.txt
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json
python bench.py --pipeline > $out/pipeline_b16.json 2> $out/pipeline.err; tail -c 3000 $out/pipeline_b16.json
