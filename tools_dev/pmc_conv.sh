#!/bin/bash
# PMC pass over one convolution layer: tools_dev/pmc_conv.sh <out_dir> "<counters>" <time_conv args...>
# (run through gpurun; rocprofv3 --pmc passes must not be combined with tracing)
out=$1; shift; ctr=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr -d $GRAFT_REPO_ROOT/gpurun_out/$out -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools_dev/time_conv.py "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$out" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:70]
    a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in agg.items():
    if "conv" not in k: continue
    print(k)
    for c, (s, n) in sorted(v.items()): print("   %-28s %14.0f per launch (%d launches)" % (c, s / n, n))
PY
