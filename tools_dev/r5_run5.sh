#!/bin/bash
# round 5, GPU run 5: retire-buffer traffic balanced over the producer waves (current library) against the committed one (libscpose_r5a.so),
# and the priority experiments of the development build (SCPOSE_DBG 128: producers without s_setprio 3; 512: consumers at priority 3 too)
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r5_run5}; mkdir -p $out
base=$root/tools_dev/ab/libscpose_r5a.so
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_hrnet.py -m gpu -x -q -k "not 2048" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
for round in 1 2 3; do
  for v in "SCPOSE_DEV=1 SCPOSE_LIB=$base" "SCPOSE_X=0"; do
    o=$(env $v python bench.py --cpu-frames 0 --steps 20 2>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'])")
    echo "[$v] $o" | tee -a $out/bench_ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
i=0
for v in "SCPOSE_DEV=1 SCPOSE_LIB=$base" "SCPOSE_X=0" "SCPOSE_DEV=1 SCPOSE_DBG=0" "SCPOSE_DEV=1 SCPOSE_DBG=128" "SCPOSE_DEV=1 SCPOSE_DBG=512"; do
  export $v
  rocprofv3 --kernel-trace --stats -d $out/t$i -o t --output-format csv -- python3 $root/bench.py --graph 0 --steps 6 --warmup 2 --cpu-frames 0 > $out/bench_t$i.json 2> $out/t$i.err
  unset SCPOSE_DEV SCPOSE_LIB SCPOSE_DBG SCPOSE_X
  i=$((i+1))
done
python3 - $out <<'PY' | tee $out/trace_ab.txt
import csv, glob, sys
out = sys.argv[1]
names = ["committed library (r5a)", "retire traffic balanced over producer waves", "development build, default", "development build, producers without s_setprio", "development build, consumers at priority 3 too"]
for m in range(5):
    f = glob.glob("%s/t%d/**/*kernel_stats.csv" % (out, m), recursive=True)
    if not f: print("no stats for", m); continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print("== variant %d: %s" % (m, names[m]))
    for r in rows[:7]:
        print("  %-72s calls %5s  avg %8.2f us  total %8.2f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $out/t0 $out/t1 $out/t2 $out/t3 $out/t4
