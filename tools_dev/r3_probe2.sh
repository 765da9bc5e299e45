#!/bin/bash
# round 3, probe 2: the new / tightened tests
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3p2; mkdir -p $out
python3 -m pytest tests/test_pnp_independent.py tests/test_gpu_pnp.py -m gpu -x -q -s > $out/pnp.txt 2>&1
python3 -m pytest tests/test_gpu_e2e.py -m gpu -x -q -s -k "every_keypoint or margin" > $out/kp.txt 2>&1
python3 -m pytest tests/test_gpu_decode.py tests/test_gpu_conv.py -m gpu -x -q > $out/dec_conv.txt 2>&1
python3 bench.py > $out/bench.json 2> $out/bench.err
tail -5 $out/*.txt
