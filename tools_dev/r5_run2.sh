#!/bin/bash
# round 5, GPU run 2: full GPU suite, then same-box bench A/B (round 4's kernels -- rebuilt with this round's ABI 7 entry point so that
# the current bench.py can drive them -- against the current library), then the pipeline side line.
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r5_run2}; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests_gpu.txt 2>&1; tail -5 $out/tests_gpu.txt
r4=$root/tools_dev/ab/libscpose_r4.so
for round in 1 2; do
  for v in "SCPOSE_DEV=1 SCPOSE_LIB=$r4" "SCPOSE_X=0"; do
    o=$(env $v python bench.py --cpu-frames 0 --steps 20 2>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'])")
    echo "[$v] $o" | tee -a $out/bench_ab.txt
  done
done
timeout 900 python bench.py --pipeline > $out/pipeline.json 2> $out/pipeline_err.txt; tail -c 1500 $out/pipeline.json; tail -5 $out/pipeline_err.txt
