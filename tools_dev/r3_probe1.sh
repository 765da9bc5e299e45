#!/bin/bash
# round 3, probe 1: baseline bench on this box, batch-size sweep of the forward, LDS / wait PMC passes
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r3p1
mkdir -p $out
cd $root
python3 bench.py > $out/bench.json 2> $out/bench.err
for n in 256 128 64; do ITERS=8 python3 tools_dev/time_forward.py w48 $n > $out/fwd_$n.txt 2>&1; done
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --steps 4 --warmup 2 --cpu-frames 0"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/pmc_a -o a --output-format csv -- $B > /dev/null 2> $out/pmc_a.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE -d $out/pmc_b -o b --output-format csv -- $B > /dev/null 2> $out/pmc_b.err
cd $root
python3 - <<'PY'
# condense the PMC CSVs on the box (they are large): per kernel name, mean of each counter per dispatch
import csv, glob, collections, os, json
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r3p1")
for tag in ("a", "b"):
    files = glob.glob(os.path.join(out, "pmc_" + tag, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    rows = {k: dict(launches=len(cnt[k]), **{c: v / len(cnt[k]) for c, v in acc[k].items()}) for k in acc}
    json.dump(rows, open(os.path.join(out, "pmc_%s_summary.json" % tag), "w"), indent=1)
    for f in files: os.remove(f)
PY
ls -la $out
