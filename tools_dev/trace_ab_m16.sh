#!/bin/bash
# Same-box, IN-FORWARD comparison of the consumer shapes: rocprofv3 kernel trace of the eager forward with SCPOSE_M16=0 and 1
# (development library), mean duration of the producer/consumer kernels.   usage (on the GPU box): tools_dev/trace_ab_m16.sh <out_dir>
root=$GRAFT_REPO_ROOT; out=$root/${1:-gpurun_out/trace_ab}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  SCPOSE_DEV=1 SCPOSE_M16=$m rocprofv3 --kernel-trace --stats -d $out/m$m -o t --output-format csv -- python3 $root/bench.py --graph 0 --steps 6 --warmup 2 --cpu-frames 0 > $out/bench_m$m.json 2> $out/m$m.err
done
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for m in (0, 1):
    f = glob.glob("%s/m%d/**/*kernel_stats.csv" % (out, m), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows if "conv" in r["Name"] or "kernel" in r["Name"])
    print("== SCPOSE_M16=%d" % m)
    for r in rows[:9]:
        print("  %-70s calls %5s  avg %8.2f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
