#!/bin/bash
# round 6, GPU run 5: the branch-chain kernel (conv_chain.hip): correctness through the W32 256x256 network tests, then configs[1] / [4]
# with and without it (SCPOSE_CHAIN=0: the per-layer path), alternating on one box
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_run5}; mkdir -p $out
lib=$root/spacecraft-pose-estimation_amd/libscpose_hip.so
timeout 1200 python -m pytest tests/test_gpu_hrnet.py -m gpu -x -q -s -k "w32_256 or captured or full_size or deterministic" > $out/tests.txt 2>&1; tail -15 $out/tests.txt
for round in 1 2 3; do
  for v in "SCPOSE_X=0" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_CHAIN=0"; do
    o=$(env $v python bench.py --model w32 --batch 64 --cpu-frames 0 --steps 30 2>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'], d['hrnet_tflops'], d['config']['launches_per_forward'])")
    echo "[w32 b64 ${v##*libscpose_hip.so}] $o" | tee -a $out/bench_ab.txt
    o=$(env $v python bench.py --events --batch 64 --cpu-frames 0 --steps 30 2>>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'], d['hrnet_tflops'])")
    echo "[events b64 ${v##*libscpose_hip.so}] $o" | tee -a $out/bench_ab.txt
  done
done
for v in "SCPOSE_X=0" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_CHAIN=0"; do
  for b in 16 256; do
    o=$(env $v python bench.py --model w32 --batch $b --cpu-frames 0 --steps 20 --graph 0 2>>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'], [(c['class'], c['avg_launch_us'], c['share_of_forward']) for c in [r]+r['next_classes']][:6])")
    echo "[w32 b$b eager ${v##*libscpose_hip.so}] $o" | tee -a $out/bench_ab.txt
  done
done
