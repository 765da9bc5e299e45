#!/bin/bash
# round 5, GPU run 3: the tests added since run 2, the direct-store experiment (SCPOSE_M16_DIRECT=1), the pipeline side line
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r5_run3}; mkdir -p $out
lib=$root/spacecraft-pose-estimation_amd/libscpose_hip.so
timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_decode.py tests/test_gpu_e2e.py -m gpu -x -q > $out/tests_new.txt 2>&1; tail -4 $out/tests_new.txt
SCPOSE_DEV=1 SCPOSE_M16_DIRECT=1 SCPOSE_LIB=$lib timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_hrnet.py -m gpu -x -q -k "not 2048" > $out/tests_direct.txt 2>&1; tail -3 $out/tests_direct.txt
for round in 1 2; do
  for v in "SCPOSE_X=0" "SCPOSE_DEV=1 SCPOSE_M16_DIRECT=1 SCPOSE_LIB=$lib"; do
    o=$(env $v python bench.py --cpu-frames 0 --steps 20 2>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'])")
    echo "[$v] $o" | tee -a $out/bench_ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
i=0
for v in "SCPOSE_X=0" "SCPOSE_DEV=1 SCPOSE_M16_DIRECT=1 SCPOSE_LIB=$lib"; do
  export $v
  rocprofv3 --kernel-trace --stats -d $out/t$i -o t --output-format csv -- python3 $root/bench.py --graph 0 --steps 6 --warmup 2 --cpu-frames 0 > $out/bench_t$i.json 2> $out/t$i.err
  unset SCPOSE_DEV SCPOSE_LIB SCPOSE_M16_DIRECT SCPOSE_X
  i=$((i+1))
done
python3 - $out <<'PY' | tee $out/trace_ab.txt
import csv, glob, sys
out = sys.argv[1]
for m in (0, 1):
    f = glob.glob("%s/t%d/**/*kernel_stats.csv" % (out, m), recursive=True)
    if not f: print("no stats for", m); continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print("== variant %d (0 = default, 1 = consumers store non-residual layers directly)" % m)
    for r in rows[:8]:
        print("  %-72s calls %5s  avg %8.2f us  total %8.2f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $out/t0 $out/t1
cd $root
timeout 1200 python bench.py --pipeline > $out/pipeline.json 2> $out/pipeline_err.txt; tail -c 1800 $out/pipeline.json; tail -3 $out/pipeline_err.txt
