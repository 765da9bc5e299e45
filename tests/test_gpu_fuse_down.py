"""fuse_down.hip: fuse row 0 + the first stride-2 hop of every down path from branch 0 of a HighResolutionModule in ONE pass
over branch 0 (landmark_regression/lib/models/pose_hrnet.py:211-239, :254-265).

The kernel forms every sum in the order of the launches it replaces (conv_s2r_kernel's k-steps; fuse_sum_kernel's j order,
fp32, one 16-bit rounding), so a forward that uses it must return the SAME BITS as one that runs the rows unfused
(development switch SCPOSE_FUSE_DOWN=0, read once per process: each arm is a sub-process).  Parity of both with the oracle is
test_gpu_hrnet.py's job; this file pins the equivalence, on shapes with partial tiles in both directions, for both branch
widths (32: two 16-channel blocks per wave; 48: three) and both storage types.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, scpose
from importlib import import_module
from oracle import hrnet_ref as R
ops = import_module('spacecraft-pose-estimation_amd.ops')
cfg = getattr(R, %(cfg)r)()
sd = R.make_state_dict(cfg, seed=11)
x = torch.randn(%(n)d, 3, %(h)d, %(w)d, generator=torch.Generator().manual_seed(12))
eng = ops.HrnetEngine(cfg, sd, dtype=%(dtype)r)
out = {'heat': eng(x.cuda()).cpu().numpy()}
for tap in %(taps)r:
    out[tap] = eng.forward_tap(x.cuda(), tap).cpu().numpy()
out['launches'] = np.array(eng.stats(%(h)d, %(w)d)['launches'])
np.savez(%(dst)r, **out)
"""

CASES = {
    "w48_96_bf16": ("w48_cfg", 2, 96, 96, "bf16"),         # branch 0: 24 x 24 -> partial tiles in x and y
    "w48_96x160_bf16": ("w48_cfg", 3, 96, 160, "bf16"),    # 24 x 40
    "w32_64_f16": ("w32_cfg", 2, 64, 64, "f16"),           # 16 x 16: one tile, three quarters of it padding
    "w32_128x192_bf16": ("w32_cfg", 2, 128, 192, "bf16"),  # 32 x 48
    "w48_256_f16": ("w48_cfg", 5, 256, 256, "f16"),        # 64 x 64: whole tiles, several per workgroup
}
TAPS = ["stage2.0.out0", "stage3.0.out0", "stage3.3.out0", "stage4.0.out0", "stage4.1.out0"]


def _run(tmp_path, name, extra_env):
    cfg, n, h, w, dtype = CASES[name]
    dst = str(tmp_path / ("%s_%s.npz" % (name, "unfused" if extra_env else "fused")))
    env = {k: v for k, v in os.environ.items() if not k.startswith("SCPOSE_")}
    env.update(extra_env)
    code = CODE % dict(root=ROOT, cfg=cfg, n=n, h=h, w=w, dtype=dtype, taps=TAPS, dst=dst)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(dst)


@pytest.mark.parametrize("name", list(CASES))
def test_fuse_down_returns_the_bits_of_the_unfused_rows(gpu_ops, tmp_path, name):
    from importlib import import_module
    nat = import_module("spacecraft-pose-estimation_amd._native")
    fused = _run(tmp_path, name, {})
    unfused = _run(tmp_path, name, {"SCPOSE_DEV": "1", "SCPOSE_FUSE_DOWN": "0", "SCPOSE_LIB": nat.LIB_PATH})
    # the switch did switch: 1 + 4 + 2 modules lose (nb - 1) stride-2 launches + the row-0 sum and gain one launch
    assert int(unfused["launches"]) - int(fused["launches"]) == 1 * 1 + 4 * 2 + 2 * 3
    for key in ["heat"] + TAPS:
        a, b = fused[key], unfused[key]
        assert a.shape == b.shape and np.isfinite(a).all()
        assert np.array_equal(a, b), "%s / %s: %d of %d elements differ, max |d| %.3e" % (
            name, key, int((a != b).sum()), a.size, float(np.abs(a - b).max()))


def test_fuse_down_frame_range_split(gpu_ops, tmp_path):
    """Tensors are addressed through 32-bit buffer descriptors, so a batch whose branch-0 tensor reaches 4 GiB (about 4 850 frames
    at 96 x 96) runs as several launches over frame ranges, every pointer advanced by its own frame size.  The development switch
    SCPOSE_FD_MAXN forces that split on five frames (ranges of 2, 2 and 1): same bits as one launch."""
    from importlib import import_module
    nat = import_module("spacecraft-pose-estimation_amd._native")
    name = "w48_256_f16"
    whole = _run(tmp_path, name, {})
    split = _run(tmp_path, name, {"SCPOSE_DEV": "1", "SCPOSE_FD_MAXN": "2", "SCPOSE_LIB": nat.LIB_PATH})
    assert int(whole["launches"]) == int(split["launches"])
    for key in ["heat"] + TAPS:
        assert np.array_equal(whole[key], split[key]), key
