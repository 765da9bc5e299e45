#!/bin/bash
# All profiler passes of one round, on the GPU box (run through gpurun):  bash tools_dev/profile_round.sh <tag>
#   headline workload (bench.py default: W48 384x384 batch 256): rocprofv3 --kernel-trace, then SEPARATE --pmc passes (never
#   combined with tracing): FETCH_SIZE, WRITE_SIZE, MFMA busy, two SQ passes (LDS / wait / issue counters);
#   side workloads (configs[1]: --model w32 --batch 64; configs[4]: --events --batch 64): kernel trace + FETCH / WRITE / MFMA.
# Each pass directory is condensed ON THE BOX by tools_dev/condense_prof.py (per kernel class, post-warm-up dispatches: median /
# p95 / mean; PMC means per launch) into gpurun_out/<tag>/<workload>/summary.json; summarise locally with
# tools_dev/summarize_round.py <tag>.
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
W=2; K=4
run_passes() {   # <name> <passes> <bench args...>
  name=$1; passes=$2; shift 2
  B="python3 $root/bench.py --steps $K --warmup $W --cpu-frames 0 --no-chain-check $*"   # (no post-run chain check under the profiler: its 64-frame forward would be attributed to the headline classes)
  d=$out/$name; mkdir -p $d
  rocprofv3 --kernel-trace -d $d/trace -o t --output-format csv -- $B > $d/bench_under_trace.json 2> $d/trace.err
  for p in $passes; do
    case $p in
      fetch) C="FETCH_SIZE";;
      write) C="WRITE_SIZE";;
      mfma)  C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE";;
      ldsa)  C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY";;
      ldsb)  C="SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS SQ_WAVES";;
    esac
    rocprofv3 --pmc $C -d $d/$p -o p --output-format csv -- $B > /dev/null 2> $d/$p.err
  done
}
# headline workload: eager launches (--graph 0) under the profiler -- the same kernels as the captured forward bench.py replays by
# default, but dispatched in launch order and one at a time, so that a dispatch can be attributed to its kernel class and its
# duration is its own (in the captured forward the fuse rows run side by side and stretch each other)
run_passes w48_b256 "fetch write mfma ldsa ldsb" --graph 0
# (side workloads too: condense_prof.py attributes dispatches to kernel classes by launch order, which a captured forward with
# concurrent lanes does not keep -- ADVICE r3)
run_passes w32_b64 "fetch write mfma" --model w32 --batch 64 --graph 0
run_passes events_b64 "fetch write mfma" --events --batch 64 --graph 0
cd $root
# one raw kernel trace survives, compressed, so that profiles/ can be re-derived from it (VERDICT r3 #6d) -- before condense_prof.py,
# which deletes the raw CSVs it has condensed
for f in $(find $out/w48_b256/trace -name "*kernel_trace.csv" | head -1); do gzip -9 -c $f > $out/w48_b256_kernel_trace.csv.gz; done
python3 tools_dev/condense_prof.py $out/w48_b256 w48 256 bf16 $W > $out/condense.log 2>&1
python3 tools_dev/condense_prof.py $out/w32_b64 w32 64 bf16 $W >> $out/condense.log 2>&1
python3 tools_dev/condense_prof.py $out/events_b64 w32 64 f16 $W >> $out/condense.log 2>&1
# the un-profiled lines
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --events --batch 64 --cpu-frames 0 > $out/bench_events.json 2>/dev/null
python3 bench.py --model w32 --batch 64 --cpu-frames 0 > $out/bench_w32_b64.json 2>/dev/null
find $out -name "*.csv" -size +1M -delete
tail -3 $out/condense.log; ls $out
