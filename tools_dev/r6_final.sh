#!/bin/bash
# round 6: the final tree -- the whole -m gpu suite, smoke(), all profiler passes (tools_dev/profile_round.sh round6_final), the driver-style bench line,
# and the pipeline at loader batch 16 and 256 on 8192 frames (enough batches at 256 for a steady-state figure)
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_final}; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests_gpu.txt 2>&1; tail -3 $out/tests_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.txt
bash tools_dev/profile_round.sh round6_final > $out/profile_round.log 2>&1; tail -2 $out/profile_round.log
python bench.py > $out/bench.json 2> $out/bench.err; python -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'], d['roofline']['class'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['chain']['kp_px_max'], d['device_state'])"
for b in 16 256; do
  python bench.py --pipeline --pipeline-frames 8192 --pipeline-batch $b > $out/pipeline_8192_b$b.json 2> $out/pipeline.err
  python -c "
import json; d=json.load(open('$out/pipeline_8192_b$b.json')); print('batch', d['batch'], d['loader_fps'], {k: (v['files_to_pred_mat_fps'], v['files_to_poses_fps']) for k, v in d['pipeline'].items()})"
done
