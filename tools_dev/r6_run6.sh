#!/bin/bash
# round 6, GPU run 6: does the captured forward overlap its lanes?  Kernel timeline of a graph replay (W32 256x256 batch 64), with and without
# the branch-chain kernel: wall, sum of kernel durations, time with 0 / 1 / >= 2 kernels in flight (tools_dev/timeline.py)
cd $GRAFT_REPO_ROOT; root=$PWD
out=$root/gpurun_out/${1:-r6_run6}; mkdir -p $out
lib=$root/spacecraft-pose-estimation_amd/libscpose_hip.so
cd /tmp && export TMPDIR=/tmp
i=0
for v in "NOTHING=0" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_CHAIN=0"; do
  export $v
  rocprofv3 --kernel-trace -d $out/t$i -o t --output-format csv -- python3 $root/tools_dev/trace_graph.py w32 64 4 > $out/trace$i.log 2> $out/trace$i.err
  unset SCPOSE_DEV SCPOSE_LIB SCPOSE_CHAIN NOTHING
  f=$(find $out/t$i -name "*kernel_trace.csv" | head -1)
  n=$([ $i = 0 ] && echo 188 || echo 258)
  echo "== variant $i ($v), last $n launches" | tee -a $out/timeline.txt
  python3 $root/tools_dev/timeline.py $f $n | tee -a $out/timeline.txt
  python3 - $f $n <<'PY' | tee -a $out/timeline.txt
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "scpose" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-int(sys.argv[2]):]
t0 = int(last[0]["Start_Timestamp"])
def short(nm): return re.sub(r"\(.*", "", nm.replace("void scpose::", "").replace("(anonymous namespace)::", ""))[:44]
# a window in the middle of the forward: the stage-4 modules
for r in last[-70:-20]:
    print("   %8.1f -> %8.1f us  q%-3s grid %6s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Queue_Id"], r["Grid_Size_X"], short(r["Kernel_Name"])))
PY
  rm -rf $out/t$i
  i=$((i+1))
done
echo "== bench A/B, captured forward (the default), W32 256x256 at batch 16 / 64 / 256" | tee -a $out/bench_graph_ab.txt
cd $root
for round in 1 2; do
  for b in 16 64 256; do
    for v in "NOTHING=0" "SCPOSE_DEV=1 SCPOSE_LIB=$lib SCPOSE_CHAIN=0"; do
      o=$(env $v python bench.py --model w32 --batch $b --cpu-frames 0 --steps 30 2>>$out/bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'], d['hrnet_tflops'], d['config']['launches_per_forward'])")
      echo "[w32 b$b ${v##*libscpose_hip.so}] $o" | tee -a $out/bench_graph_ab.txt
    done
  done
done
