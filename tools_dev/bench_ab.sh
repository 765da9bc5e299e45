#!/bin/bash
# same-box comparison of bench.py variants: tools_dev/bench_ab.sh "<env/args A>" "<env/args B>" ...  (each: "VAR=.. VAR=.. -- --flag ..")
cd "$(dirname "$0")/.."
for round in 1 2; do
  for v in "$@"; do
    envs="${v%%--*}"; args="${v#*--}"; [ "$args" == "$v" ] && args=""
    out=$(env $envs python bench.py --cpu-frames 0 --steps 20 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hrnet_forward_ms'])")
    echo "[$v] $out"
  done
done
