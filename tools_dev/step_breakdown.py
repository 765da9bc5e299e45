"""Where a bench.py step's time beyond its forward goes, in one process on one box:
   (a) forward alone, (b) forward + decode + PnP serially on one stream, (c) the pipelined step of bench.py
   (decode / PnP / D2H of step i on a side stream beside the forward of step i+1), (d) as (c) without PnP, (e) as (c) without decode.
usage: python tools_dev/step_breakdown.py [N]"""
import _dev  # noqa: F401
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scpose
from importlib import import_module
ops = import_module("spacecraft-pose-estimation_amd.ops")
syn = import_module("spacecraft-pose-estimation_amd.synthetic")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
image, J = 384, 11
cfg = syn.hrnet_cfg(48, J, image)
eng = ops.HrnetEngine(cfg, syn.random_checkpoint(cfg, 0))
dev = torch.device("cuda")
frames = syn.rgb_crops(B, image, torch.Generator().manual_seed(1)).to(dev)
center = torch.full((B, 2), image / 2.0, device=dev); scale = torch.full((B, 2), image / 200.0 * 1.5, device=dev)
kp_np, _, _ = syn.keypoints(B, np.random.default_rng(2), noise_px=1.0, outlier_frac=0.1)
kp = torch.from_numpy(kp_np).to(dev)
lm, Kc, dc = (torch.from_numpy(a).to(dev) for a in (syn.TANGO_LANDMARKS, syn.SPEEDPLUS_K, syn.SPEEDPLUS_DIST))
heat = [torch.empty((B, J, 96, 96), device=dev) for _ in range(2)]
host = [torch.empty((B, 13), dtype=torch.float64).pin_memory() for _ in range(2)]
side = torch.cuda.Stream()

def post(k, do_dec=True, do_pnp=True):
    if do_dec: ops.decode(heat[k], center, scale, True)
    if do_pnp:
        rot, tv, st = ops.pnp_epnp_ransac(kp, lm, Kc, dc)
        blk = torch.cat([rot.view(B, 9), tv, st.double().unsqueeze(1)], 1)
        host[k].copy_(blk, non_blocking=True)

def run(mode, iters=12):
    done = [None, None]
    def one(i):
        k = i & 1
        main = torch.cuda.current_stream()
        if mode == "fwd":
            eng.forward(frames, out=heat[k]); return
        if mode == "serial":
            eng.forward(frames, out=heat[k]); post(k); return
        if done[k] is not None: main.wait_event(done[k])
        eng.forward(frames, out=heat[k])
        ev = torch.cuda.Event(); ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            post(k, do_dec=mode != "pipe_nodec", do_pnp=mode != "pipe_nopnp")
            done[k] = torch.cuda.Event(); done[k].record(side)
    for i in range(3): one(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(iters): one(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3

MODES = os.environ.get("MODES", "fwd,serial,pipe,pipe_nopnp,pipe_nodec").split(",")
for rep in range(2):
    print("  ".join("%s %.3f ms" % (m, run(m)) for m in MODES))
