#!/bin/bash
# All profiler passes of one round, on the GPU box (run through gpurun):  tools_dev/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of bench.py (the command whose JSON line is the round's figure)
#   2.-4. separate --pmc passes (never combined with tracing): FETCH_SIZE, WRITE_SIZE, MFMA busy
# Raw output under gpurun_out/<tag>_{stats,fetch,write,mfma}; summarise locally with tools_dev/summarize_round.py <tag>.
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --steps 4 --warmup 2 --cpu-frames 0"
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/${tag}_stats -o s --output-format csv -- $B > $root/gpurun_out/${tag}_bench_under_prof.json 2> $root/gpurun_out/${tag}_stats.err
rocprofv3 --pmc FETCH_SIZE -d $root/gpurun_out/${tag}_fetch -o f --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $root/gpurun_out/${tag}_write -o w --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $root/gpurun_out/${tag}_mfma -o m --output-format csv -- $B > /dev/null 2>&1
cd $root
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 bench.py --events --batch 64 --cpu-frames 0 > gpurun_out/${tag}_bench_events.json 2>/dev/null
python3 bench.py --model w32 --batch 64 --cpu-frames 0 > gpurun_out/${tag}_bench_w32_b64.json 2>/dev/null
rm -f gpurun_out/${tag}_stats/*kernel_trace.csv    # large; the stats file is what gets summarised
ls gpurun_out/${tag}_*
