"""The fused tail of pose_hrnet (csrc/head_fused.hip): last fuse sum + final_layer (+ decode) in one pass.

What the reference does in three steps -- HighResolutionModule's last fuse row (landmark_regression/lib/models/pose_hrnet.py:256-263),
final_layer (:458) and get_final_preds (lib/core/inference.py:49-79, called from lib/core/function.py:376-393) -- is one kernel
here.  The whole-network parity tests (test_gpu_hrnet.py) already run through it; this file pins what is specific to the fusion:
  * heat-maps: equal to final_layer applied to the un-fused fuse-row output (the `stage<S>.<M>.out0` tap) up to fp32 summation order;
  * key points: scpose_hrnet_forward_decode is BIT-identical to scpose_decode(scpose_hrnet_forward(...)), with and without the
    heat-map buffer, eager and captured, including all-equal maps (first index wins), non-positive maxima (masked) and NaN maps;
  * maps whose width is not a multiple of the 16-pixel MFMA column.
"""
import numpy as np
import pytest
import torch

from oracle import hrnet_ref as R

pytestmark = pytest.mark.gpu

CASES = {
    "tiny96x64_n3": (R.tiny_cfg(), (96, 64), 3),      # 24 x 16 maps
    "w32_288x224_n2": (R.w32_cfg(), (288, 224), 2),   # 72 x 56 maps: columns of 16 pixels straddle rows
    "w32_256_n5": (R.w32_cfg(), (256, 256), 5),
    "w48_384_n2": (R.w48_cfg(), (384, 384), 2),       # BASELINE config B geometry
}


def _inputs(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, generator=g)
    center = torch.rand(n, 2, generator=g) * 400 + 300
    scale = torch.rand(n, 2, generator=g) * 2 + 1
    return x.cuda(), center.cuda(), scale.cuda()


def _last_tap(eng):
    taps = [t for t in eng.tap_names() if t.endswith(".out0")]
    return taps[-1]


@pytest.mark.parametrize("name", list(CASES))
def test_fused_tail_heatmaps_equal_final_layer_of_the_unfused_fuse_row(gpu_ops, name):
    cfg, (h, w), n = CASES[name]
    sd = R.make_state_dict(cfg, seed=11)
    eng = gpu_ops.HrnetEngine(cfg, sd, dtype="bf16")
    assert eng.tail_fused(n, h, w)
    x, _, _ = _inputs(n, h, w, 1)
    hm = eng(x)
    y0 = eng.forward_tap(x, _last_tap(eng))           # stops at the fuse row: fuse_sum runs as its own kernel, 16-bit values as f32
    wf = sd["final_layer.weight"].float().reshape(hm.shape[1], -1).bfloat16().float().cuda()
    ref = torch.einsum("jc,nchw->njhw", wf.double(), y0.double()) + sd["final_layer.bias"].double().cuda()[None, :, None, None]
    err = (hm.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print("%s: max |fused - final_layer(tap)| = %.3e (max |hm| %.3e)" % (name, err, scale))
    assert err <= 2e-6 * max(scale, 1.0) + 1e-6       # fp32 accumulation of 32-64 exact products: summation order only
    eng.close()


@pytest.mark.parametrize("post", [True, False])
@pytest.mark.parametrize("name", list(CASES))
def test_forward_decode_is_bit_identical_to_forward_then_decode(gpu_ops, name, post):
    cfg, (h, w), n = CASES[name]
    sd = R.make_state_dict(cfg, seed=12)
    eng = gpu_ops.HrnetEngine(cfg, sd, dtype="bf16")
    x, center, scale = _inputs(n, h, w, 2)
    hm = eng(x)
    want = gpu_ops.decode(hm, center, scale, post)
    got = eng.forward_decode(x, center, scale, post)                       # no heat-map buffer at all
    got2, hm2 = eng.forward_decode(x, center, scale, post, heatmaps=True)
    assert got.shape == (n, hm.shape[1], 3)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    assert torch.equal(got2.view(torch.int32), want.view(torch.int32))
    assert torch.equal(hm2.view(torch.int32), hm.view(torch.int32))
    # and the quarter-pixel branch was actually exercised
    if post:
        plain = gpu_ops.decode(hm, center, scale, False)
        assert not torch.equal(plain, want)
    eng.close()


def test_forward_decode_ties_masked_maxima_and_nan_maps(gpu_ops):
    """final_layer with zero weights: every map is constant (its bias) -- the first pixel wins; a non-positive maximum masks the
    coordinates (inference.py:41-45); a NaN bias makes the whole map NaN (np.argmax: index 0, maxval NaN)."""
    cfg = R.tiny_cfg()
    sd = R.make_state_dict(cfg, seed=13)
    sd["final_layer.weight"] = torch.zeros_like(sd["final_layer.weight"])
    b = torch.linspace(-1.0, 1.0, sd["final_layer.bias"].numel())
    b[3] = float("nan")
    b[5] = 0.0
    sd["final_layer.bias"] = b
    eng = gpu_ops.HrnetEngine(cfg, sd, dtype="bf16")
    x, center, scale = _inputs(4, 64, 96, 3)
    hm = eng(x)
    assert torch.isnan(hm[:, 3]).all() and (hm[:, 5] == 0).all()
    want = gpu_ops.decode(hm, center, scale, True)
    got = eng.forward_decode(x, center, scale, True)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    assert torch.isnan(got[:, 3, 2]).all()
    eng.close()

    # a map with ONE distinct maximum per joint, placed by the weights of a one-hot-ish network is covered by the random
    # cases above; here, additionally, exact ties between distant pixels: weights zero, so all 24 x 16 pixels tie
    coords, _ = gpu_ops.max_preds(hm)
    assert (coords[:, [0, 1, 2, 4]] == 0).all()


@pytest.mark.parametrize("concurrent", [0, 1, 2])
def test_captured_forward_decode_replays_bit_identically(gpu_ops, concurrent):
    cfg = R.w32_cfg()
    sd = R.make_state_dict(cfg, seed=14)
    eng = gpu_ops.HrnetEngine(cfg, sd, dtype="bf16")
    n, h, w = 3, 128, 128
    x, center, scale = _inputs(n, h, w, 4)
    want = eng.forward_decode(x, center, scale, True)
    g = eng.capture_decode(x, center, scale, True, concurrent=concurrent)
    assert g.heatmaps is None
    got = g.replay().clone()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    x2, c2, s2 = _inputs(n, h, w, 5)
    x.copy_(x2); center.copy_(c2); scale.copy_(s2)     # bound by address: refill in place
    got = g.replay().clone()
    want = eng.forward_decode(x, center, scale, True)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    g.close()
    eng.close()


def test_profile_reports_the_fused_tail_as_its_own_kernel_class(gpu_ops):
    cfg = R.w32_cfg()
    eng = gpu_ops.HrnetEngine(cfg, R.make_state_dict(cfg, seed=15), dtype="bf16")
    x, _, _ = _inputs(2, 128, 128, 6)
    eng.forward(x, profile=True)
    recs = eng.profile_read()
    kinds = [r["kind"] for r in recs]
    assert kinds.count(7) == 1 and kinds[-1] == 7
    tail = recs[-1]
    assert tail["cout"] == 11 and tail["flops_per_frame"] == 2.0 * tail["cin"] * 11 * 32 * 32
    absorbed = recs[-2]                                 # the fuse row the tail absorbs: launched nothing
    assert absorbed["kind"] == 2 and absorbed["flops_per_frame"] == 0 and absorbed["bytes_per_frame"] == 0
    eng.close()


def test_unfusable_tail_still_decodes_through_a_heatmap_buffer(gpu_ops):
    """A 3x3 final layer (like the hrnet_cms heads) has no fused tail: forward_decode = forward + decode on the same stream."""
    cfg = R.tiny_cfg()
    cfg = dict(cfg)
    cfg["MODEL"] = dict(cfg["MODEL"]); cfg["MODEL"]["EXTRA"] = dict(cfg["MODEL"]["EXTRA"]); cfg["MODEL"]["EXTRA"]["FINAL_CONV_KERNEL"] = 3
    sd = R.make_state_dict(cfg, seed=16)
    eng = gpu_ops.HrnetEngine(cfg, sd, dtype="bf16")
    assert not eng.tail_fused(2, 64, 64)
    x, center, scale = _inputs(2, 64, 64, 7)
    want = gpu_ops.decode(eng(x), center, scale, True)
    got = eng.forward_decode(x, center, scale, True)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    eng.close()
