#!/bin/bash
# round 6: what the driver runs at round end, on the final tree -- the GPU suite, smoke(), the default bench line
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_final2}; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -x -q > $out/tests_gpu.txt 2>&1; tail -3 $out/tests_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.txt
python bench.py > $out/bench.json 2> $out/bench.err; python -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'], d['roofline']['class'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['chain']['kp_px_max'], d['step_ms'], d['device_state']['sclk_mhz_median'])"
