#!/bin/bash
# round 6, GPU run 1: go / no-go of TWO consumer waves per SIMD (conv_m32p_kernel.h, CW2) in the real kernel, same box:
#   SCPOSE_M16_CW2 = 0 (product default) | 2 (layers with several Cout blocks: 192 / 384 channels) | 1 (96 -> 96 too, weights by LDS-DMA)
#   and the pure A/B on the 256-pixel tile group: SCPOSE_M16_NB=4 with and without CW2 (4 consumer waves x 6 x 4 against 8 x 6 x 2)
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_run1}; mkdir -p $out
lib=$root/spacecraft-pose-estimation_amd/libscpose_hip.so
D="SCPOSE_DEV=1 SCPOSE_LIB=$lib"
echo "== correctness of the CW2 variants (conv tests under the switch)" | tee $out/tests.txt
for v in "SCPOSE_M16_CW2=1" "SCPOSE_M16_CW2=1 SCPOSE_M16_NB=4"; do
  env $D $v timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -3 | tee -a $out/tests.txt
done
env $D SCPOSE_M16_CW2=1 timeout 900 python -m pytest tests/test_gpu_hrnet.py -m gpu -x -q -k "not 2048" 2>&1 | tail -3 | tee -a $out/tests.txt
echo "== bench A/B, alternating" | tee $out/bench_ab.txt
for round in 1 2 3; do
  for v in "SCPOSE_X=0" "$D SCPOSE_M16_CW2=2" "$D SCPOSE_M16_CW2=1" "$D SCPOSE_M32_WREG=0"; do
    o=$(env $v python bench.py --cpu-frames 0 --steps 20 2>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'])")
    echo "[${v##*libscpose_hip.so}] $o" | tee -a $out/bench_ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
i=0
names=("default" "CW2=2" "CW2=1" "WREG=0" "NB=4" "NB=4 CW2=1")
for v in "SCPOSE_X=0" "$D SCPOSE_M16_CW2=2" "$D SCPOSE_M16_CW2=1" "$D SCPOSE_M32_WREG=0" "$D SCPOSE_M16_NB=4" "$D SCPOSE_M16_NB=4 SCPOSE_M16_CW2=1"; do
  export $v
  rocprofv3 --kernel-trace --stats -d $out/t$i -o t --output-format csv -- python3 $root/bench.py --graph 0 --steps 6 --warmup 2 --cpu-frames 0 > $out/bench_t$i.json 2> $out/t$i.err
  unset SCPOSE_DEV SCPOSE_LIB SCPOSE_M16_CW2 SCPOSE_M32_WREG SCPOSE_M16_NB SCPOSE_X
  i=$((i+1))
done
python3 - $out <<'PY' | tee $out/trace_ab.txt
import csv, glob, sys
out = sys.argv[1]
names = ["default", "CW2=2 (192 / 384 channels)", "CW2=1 (all 96-row stride-1 layers; 96->96 weights by LDS-DMA)", "WREG=0 (96->96 weights by LDS-DMA, one consumer wave per SIMD)", "NB=4 (256-pixel tile groups, 4 x 6x4)", "NB=4 CW2=1 (256-pixel tile groups, 8 x 6x2)"]
for m in range(6):
    f = glob.glob("%s/t%d/**/*kernel_stats.csv" % (out, m), recursive=True)
    if not f: print("no stats for", m); continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print("== variant %d: %s" % (m, names[m]))
    for r in rows[:8]:
        print("  %-76s calls %5s  avg %8.2f us  total %8.2f ms" % (r["Name"][:76], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $out/t0 $out/t1 $out/t2 $out/t3 $out/t4 $out/t5
