#!/bin/bash
# round 5, experiment 1 (on the GPU box): one addressing mode per kernel (ADDR) and the barrier-free schedule (SCPOSE_M16_SYNC=1)
# against round 4's library -- correctness first, then same-box bench.py A/B and an in-forward kernel trace per setting.
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r5_exp1}; mkdir -p $out
r4=$root/tools_dev/ab/libscpose_r4.so
SCPOSE_DEV=1 SCPOSE_M16_SYNC=1 SCPOSE_LIB=$root/spacecraft-pose-estimation_amd/libscpose_hip.so timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_hrnet.py -m gpu -x -q > $out/tests_sync.txt 2>&1; tail -3 $out/tests_sync.txt
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q > $out/tests_e1.txt 2>&1; tail -1 $out/tests_e1.txt
for round in 1 2; do
  for v in "SCPOSE_DEV=1 SCPOSE_LIB=$r4" "SCPOSE_X=0" "SCPOSE_DEV=1 SCPOSE_M16_SYNC=1 SCPOSE_LIB=$root/spacecraft-pose-estimation_amd/libscpose_hip.so"; do
    o=$(env $v python bench.py --cpu-frames 0 --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'])")
    echo "[$v] $o" | tee -a $out/bench_ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
i=0
for v in "SCPOSE_DEV=1 SCPOSE_LIB=$r4" "SCPOSE_X=0" "SCPOSE_DEV=1 SCPOSE_M16_SYNC=1 SCPOSE_LIB=$root/spacecraft-pose-estimation_amd/libscpose_hip.so"; do
  export $v
  rocprofv3 --kernel-trace --stats -d $out/t$i -o t --output-format csv -- python3 $root/bench.py --graph 0 --steps 6 --warmup 2 --cpu-frames 0 > $out/bench_t$i.json 2> $out/t$i.err
  unset SCPOSE_DEV SCPOSE_LIB SCPOSE_M16_SYNC SCPOSE_X
  i=$((i+1))
done
python3 - $out <<'PY' | tee $out/trace_ab.txt
import csv, glob, sys
out = sys.argv[1]
for m in (0, 1, 2):
    f = glob.glob("%s/t%d/**/*kernel_stats.csv" % (out, m), recursive=True)
    if not f: print("no stats for", m); continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print("== variant %d (0 = round 4, 1 = ADDR split, 2 = ADDR + SYNC)" % m)
    for r in rows[:8]:
        print("  %-72s calls %5s  avg %8.2f us  total %8.2f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $out/t0 $out/t1 $out/t2
