#!/bin/bash
# round 6, GPU run 4: all profiler passes of the round (tools_dev/profile_round.sh), the pipeline at loader batch 256 beside run 3's batch 16,
# and the default bench line again (device_state through the PCI address)
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_run4}; mkdir -p $out
bash tools_dev/profile_round.sh round6_final > $out/profile_round.log 2>&1; tail -3 $out/profile_round.log
python bench.py --pipeline --pipeline-batch 256 > $out/pipeline_b256.json 2> $out/pipeline.err; tail -c 1200 $out/pipeline_b256.json
python bench.py --steps 20 > $out/bench_steps20.json 2> $out/bench.err; python -c "
import json; d=json.load(open('$out/bench_steps20.json')); print(d['value'], d['ms_per_step'], d['step_ms'], d['device_state'], d['roofline']['traffic'])"
