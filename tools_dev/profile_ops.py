"""Per-launch-class timing of the HIP forward via scpose_hrnet_forward_profiled.
usage: python tools_dev/profile_ops.py [w48|w32] [N] [size]"""
import _dev  # noqa: F401  (enables the library's development switches when SCPOSE_* variables are set)
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scpose
from importlib import import_module
ops = import_module("spacecraft-pose-estimation_amd.ops")
syn = import_module("spacecraft-pose-estimation_amd.synthetic")
which = sys.argv[1] if len(sys.argv) > 1 else "w48"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
size = int(sys.argv[3]) if len(sys.argv) > 3 else (384 if which == "w48" else 256)
cfg = syn.hrnet_cfg(48 if which == "w48" else 32, 11, size)
cfg["MODEL"]["NAME"] = os.environ.get("SCPOSE_MODEL", "pose_hrnet")
eng = ops.HrnetEngine(cfg, syn.random_checkpoint(cfg, 0))
x = torch.randint(0, 256, (n, size, size, 3), dtype=torch.uint8, device="cuda")
for _ in range(2): eng(x)
agg = collections.OrderedDict()
reps = 3
for _ in range(reps):
    eng.forward(x, profile=True)
    for i, r in enumerate(eng.profile_read()):
        hw = int(round((r["bytes_per_frame"]) ))
        key = (r["kind"], r["a"], r["cin"], r["cout"], int(r["flops_per_frame"]))
        e = agg.setdefault(key, [0.0, 0, r["flops_per_frame"] * n, r["bytes_per_frame"] * n])
        e[0] += r["ms"]; e[1] += 1
tot = sum(v[0] for v in agg.values()) / reps
print("total %.2f ms" % tot)
print("%-28s %5s %9s %8s %8s %8s %6s" % ("op", "calls", "avg_us", "tot_ms", "TFLOP/s", "GB/s", "%"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    kind, a, cin, cout, fl = k
    name = {0: "stem", 2: "fuse%d C%d" % (a, cin), 3: "block C%d (2x k3 s1)" % cin,
            4: "head gather k%d s%d %d->%d" % (a // 10, a % 10, cin, cout)}.get(kind, "conv k%d s%d %d->%d" % (a // 10, a % 10, cin, cout))
    calls = v[1] // reps
    avg = v[0] / v[1]
    print("%-28s %5d %9.1f %8.2f %8.1f %8.0f %6.1f" % (name + " f%.0fM" % (fl / 1e6), calls, avg * 1e3, v[0] / reps, v[2] / avg / 1e9, v[3] / avg / 1e6, 100 * v[0] / reps / tot))
