#!/bin/bash
# tile choices of the 32x32 / 16x16x32 producer-consumer layers at small batches (SCPOSE_DBG=32 prints them)
cd $GRAFT_REPO_ROOT; root=$PWD
out=gpurun_out/${1:-r5_quick}; mkdir -p $out
lib=$root/spacecraft-pose-estimation_amd/libscpose_hip.so
for b in 32 64 256; do
  SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_DBG=32 python bench.py --batch $b --steps 1 --warmup 0 --cpu-frames 0 --graph 0 2>&1 >/dev/null | grep "^m32" | sort | uniq -c | sort -rn > $out/tiles_b$b.txt
  echo "== batch $b"; head -12 $out/tiles_b$b.txt
done
