#!/bin/bash
# Board power and shader clock while the forward runs (rocm-smi polled from a second process): tools_dev/power_trace.sh
cd $GRAFT_REPO_ROOT
rocm-smi --showpower --showclocks --showperflevel 2>&1 | grep -v "^$" | head -30
echo "=== under load (W48 384^2 batch 256 forward loop)"
ITERS=1200 python3 tools_dev/time_forward.py w48 256 > gpurun_out/power_fwd.log 2>&1 &
PID=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | tr '\n' ';'; echo
  sleep 1
done
wait $PID
tail -1 gpurun_out/power_fwd.log
echo "=== idle again"
sleep 3
rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | tr '\n' ';'; echo
rocm-smi --showmaxpower 2>&1 | grep -i power
