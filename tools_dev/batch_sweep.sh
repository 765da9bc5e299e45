#!/bin/bash
# Batch sweep of the headline workload (W48 384x384 + PnP) on one box: VERDICT r3 #6a.
#   tools_dev/batch_sweep.sh <out_dir> [batches...]      (on the GPU box; condensed by tools_dev/summarize_sweep.py)
out=${1:-gpurun_out/sweep}; shift
batches=${@:-"32 64 128 256 512"}
mkdir -p "$out"
for b in $batches; do
  python3 bench.py --batch "$b" --steps 12 --warmup 4 --cpu-frames 0 > "$out/bench_b$b.json" 2> "$out/bench_b$b.err" || echo "batch $b failed" >&2
done
python3 - "$out" $batches <<'PY'
import json, sys
out, batches = sys.argv[1], [int(b) for b in sys.argv[2:]]
rows = []
for b in batches:
    try:
        j = json.loads(open("%s/bench_b%d.json" % (out, b)).read().strip().splitlines()[-1])
    except Exception as e:
        rows.append({"batch": b, "error": str(e)}); continue
    rows.append({"batch": b, "poses_per_s": j["value"], "ms_per_step": j["ms_per_step"], "forward_ms": j["hrnet_forward_ms"],
                 "forward_us_per_frame": round(j["hrnet_forward_ms"] * 1e3 / b, 2), "step_us_per_frame": round(j["ms_per_step"] * 1e3 / b, 2),
                 "hrnet_tflops": j["hrnet_tflops"], "dominant": j["roofline"]["class"], "dominant_us": j["roofline"]["avg_launch_us"]})
json.dump({"workload": "HRNet-W48 384x384 11 joints + EPnP-RANSAC, one MI355X, same box, bench.py --batch B --steps 12 --warmup 4", "rows": rows},
          open(out + "/batch_sweep.json", "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
