#!/bin/bash
# round 5, GPU run 6: the library with the priority change against the committed r5a build; small-batch sweeps of the tile-search switches
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r5_run6}; mkdir -p $out
base=$root/tools_dev/ab/libscpose_r5a.so
lib=$root/spacecraft-pose-estimation_amd/libscpose_hip.so
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q > $out/tests.txt 2>&1; tail -2 $out/tests.txt
for round in 1 2 3; do
  for v in "SCPOSE_DEV=1 SCPOSE_LIB=$base" "SCPOSE_X=0"; do
    o=$(env $v python bench.py --cpu-frames 0 --steps 20 2>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'])")
    echo "[$v] $o" | tee -a $out/bench_ab.txt
  done
done
echo "== W32 256x256 batch 64 (configs[1]) under tile-search switches" | tee -a $out/small.txt
for v in "SCPOSE_X=0" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_CPMUL=1" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_CPMUL=2" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_CUS=64" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_CUS=128" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_OCC=2" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_OCC=3" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_NR=1" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_K1_OCC=1"; do
  o=$(env $v python bench.py --model w32 --batch 64 --cpu-frames 0 --steps 30 2>>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'])")
  echo "[${v##*libscpose_hip.so}] $o" | tee -a $out/small.txt
done
echo "== W48 384x384 batch 32 under the same switches" | tee -a $out/small.txt
for v in "SCPOSE_X=0" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_CPMUL=1" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_CPMUL=2" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M32_CUS=128" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M16_NB=4" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_M16_NB=2"; do
  o=$(env $v python bench.py --batch 32 --cpu-frames 0 --steps 30 2>>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'], d['hrnet_tflops'])")
  echo "[${v##*libscpose_hip.so}] $o" | tee -a $out/small.txt
done
