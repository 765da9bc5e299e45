#!/bin/bash
# quick same-box sweeps of existing switches on the final library (bench.py --steps 20: poses/s, ms per step, forward ms)
cd $GRAFT_REPO_ROOT; root=$PWD
out=gpurun_out/${1:-r5_quick}; mkdir -p $out
lib=$root/spacecraft-pose-estimation_amd/libscpose_hip.so
for round in 1 2; do
for v in "SCPOSE_X=0" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_X=1" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_WREG=0" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_NST=1" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_NST=3" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_TILE=16,24" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M16_NB=5" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M16_NB=4"; do
  o=$(env $v python bench.py --cpu-frames 0 --steps 20 2>>$out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'])")
  echo "[${v##*libscpose_hip.so}] $o" | tee -a $out/sweep.txt
done
done
