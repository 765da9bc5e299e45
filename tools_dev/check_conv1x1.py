"""Determinism of the streaming 1x1 kernel per K-step variant (developer diagnostic)."""
import _dev  # noqa: F401
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, scpose
from importlib import import_module
ops = import_module("spacecraft-pose-estimation_amd.ops")
g = torch.Generator().manual_seed(0)
for cin, cout, H, N in ((32, 400, 64, 24), (64, 256, 96, 32), (96, 48, 48, 64), (128, 64, 32, 64), (192, 96, 24, 64), (256, 64, 96, 32), (384, 48, 12, 64), (256, 32, 16, 64)):
    conv = ops.Conv(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, torch.randn(cout, generator=g) * 0.1)
    x = torch.randn(N, cin // 8, H, H, 8, generator=g).bfloat16().cuda()
    ys = [conv(x, relu=True).float() for _ in range(4)]
    bad = [int((ys[i] != ys[0]).sum()) for i in range(1, 4)]
    print("1x1 %3d->%3d %dx%d N=%d ksteps %d: %s" % (cin, cout, H, H, N, (cin // 8 + 3) // 4, "deterministic" if not any(bad) else "DIFFERS %s" % bad))
