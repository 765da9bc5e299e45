#!/bin/bash
# round 6, GPU run 2: what bounds the 3x3 s1 producer/consumer kernel?  Ablations of the development build on isolated layers
# (SCPOSE_DBG bits: 1 = no MFMA loops, 2 = no retire-buffer traffic (residual loads + stores), 4 = no halo DMA), NB=5 vs 6, CW2.
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_run2}; mkdir -p $out
for shape in "96 96 3 1 48 256 res" "96 96 3 1 48 256" "192 192 3 1 24 256 res" "384 384 3 1 12 256 res"; do
  echo "== $shape" | tee -a $out/ablate.txt
  for dbg in 0 1 2 4 6 3 5 7; do
    SCPOSE_DEV=1 SCPOSE_DBG=$dbg ITERS=100 python3 tools_dev/time_conv.py $shape 2>/dev/null | tail -1 | tee -a $out/ablate.txt
  done
  for v in "SCPOSE_M16_NB=5" "SCPOSE_M16_NB=5 SCPOSE_DBG=1" "SCPOSE_M16_CW2=1" "SCPOSE_M16_CW2=1 SCPOSE_DBG=1" "SCPOSE_M32_WREG=0" "SCPOSE_NST=1" "SCPOSE_NST=3"; do
    echo -n "[$v] " | tee -a $out/ablate.txt
    env SCPOSE_DEV=1 $v ITERS=100 python3 tools_dev/time_conv.py $shape 2>/dev/null | tail -1 | tee -a $out/ablate.txt
  done
done
