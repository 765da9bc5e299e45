#!/bin/bash
# round 5, GPU run 4: full GPU suite on the final kernels, pipeline side line (forkserver and fork workers), then all profiler passes
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r5_run4}; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests_gpu.txt 2>&1; tail -4 $out/tests_gpu.txt
timeout 900 python bench.py --pipeline > $out/pipeline_forkserver.json 2> $out/pipeline_err.txt; tail -c 1600 $out/pipeline_forkserver.json
timeout 900 python tools_dev/pipeline_bench.py --mp fork > $out/pipeline_fork.json 2>> $out/pipeline_err.txt; tail -c 1200 $out/pipeline_fork.json
timeout 600 python bench.py --fitted-w48 --batch 256 --cpu-frames 0 > $out/bench_fitted_w48.json 2> $out/bench_fitted_err.txt; python -c "
import json; d=json.load(open('$out/bench_fitted_w48.json')); print(d['value'], d['ms_per_step'], d['chain'], d['poses_ok'])"
bash tools_dev/profile_round.sh ${1:-r5_run4}/prof > $out/profile_round.log 2>&1; tail -3 $out/profile_round.log
