#!/bin/bash
# round 5, GPU run 7: all profiler passes on the final kernels, the pipeline side line with pre-loaded fork-server workers, the CLI tests
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r5_run7}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_cms.py -m gpu -x -q > $out/tests_cli.txt 2>&1; tail -3 $out/tests_cli.txt
timeout 900 python bench.py --pipeline > $out/pipeline.json 2> $out/pipeline_err.txt; tail -c 1900 $out/pipeline.json
bash tools_dev/profile_round.sh round5_final > $out/profile_round.log 2>&1; tail -3 $out/profile_round.log
