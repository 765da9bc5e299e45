#!/bin/bash
# round 6, GPU run 11: the fused C = 48 BasicBlock with the input tile's LDS-DMA issued by the conv1 waves (SCPOSE_BLOCK_DMA=0) instead of the conv2 waves
# (SCPOSE_BLOCK_DMA was an experimental template switch of conv_block2_kernel.h, measured here and NOT kept in the sources: profiles/round6_block_dma_role_ab.txt)
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_run11}; mkdir -p $out
lib=$root/spacecraft-pose-estimation_amd/libscpose_hip.so
D="SCPOSE_DEV=1 SCPOSE_LIB=$lib"
env $D SCPOSE_BLOCK_DMA=0 timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "block" 2>&1 | tail -2 | tee $out/tests.txt
env $D SCPOSE_BLOCK_DMA=0 timeout 900 python -m pytest tests/test_gpu_hrnet.py -m gpu -x -q -k "w48_384 or w32_256 or captured or deterministic" 2>&1 | tail -2 | tee -a $out/tests.txt
for d in relu; do
  for v in 1 0; do
    echo "== stamps SCPOSE_BLOCK_DMA=$v (development build, isolated)" | tee -a $out/stamps.txt
    SCPOSE_DEV=1 SCPOSE_DBG=8 SCPOSE_BLOCK_DMA=$v ITERS=50 python3 tools_dev/time_block.py 48 96 256 $d 2>&1 | tail -11 | tee -a $out/stamps.txt
  done
done
for round in 1 2 3; do
  for v in "$D SCPOSE_BLOCK_DMA=1" "$D SCPOSE_BLOCK_DMA=0"; do
    o=$(env $v python bench.py --cpu-frames 0 --steps 20 --no-chain-check 2>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'], r['class'], r['avg_launch_us'])")
    echo "[${v##*libscpose_hip.so}] $o" | tee -a $out/bench_ab.txt
  done
done
