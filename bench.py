#!/usr/bin/env python3
"""End-to-end poses/s of the HIP HRNet -> decode -> EPnP/RANSAC path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N>1 either under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the driver's
    launch: RANK / LOCAL_RANK / WORLD_SIZE come from the environment) or stand-alone: without WORLD_SIZE in the
    environment `bench.py --gpus N` starts its N ranks itself as fresh child processes (parallel.spawn_local_ranks,
    before anything touches the GPU in the parent) and exits non-zero if one of them dies.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): HRNet-W48,
384x384 synthetic RGB crops, 11 landmarks, 256 frames per GPU per step, bf16 MFMA; weak scaling:
every rank processes its own contiguous shard of the frame list and the per-rank (R, t,
status) blocks are all-gathered over RCCL (13 float64 per frame) -- SURVEY.md section 8(e).
At N = 8 this is BASELINE.json configs[3]: 2048 frames per step, 256 per rank.
`--events` switches to configs[4] (side line, not the headline): HRNet-W32 256x256 on the f16 MFMA
kernels, every batch half RGB noise crops and half synthetic event frames.

A step, timed with inputs resident in HBM: uint8 crops -> pose_hrnet forward -> heatmap decode
-> batched EPnP+RANSAC -> all-gather -> (R, t, status) on the host of rank 0.
The random-init network's heatmaps carry no pose, so -- as SURVEY.md section 8(d) prescribes --
the PnP stage consumes seeded synthetic keypoints (landmarks projected through random poses,
1 px noise, 10 % outliers) of the same shape; the decode stage still runs on the network's
heatmaps inside the timed region.  Nothing is skipped or cached between steps.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel class (largest summed time), from HIP events recorded around every
                launch (scpose_hrnet_forward_profiled, on the launch stream) of PROF_STEPS extra steps that run
                right after the timed region -- the K timed steps themselves carry no per-launch events, only
                one event pair around each forward (`hrnet_forward_ms`).  Algorithmic bytes
                are the bytes THAT launch has to move (a fused BasicBlock: its input once + its output
                once), the bound is chosen from flops / those bytes against the ridge, and `frac` is
                against that bound's peak.  `traffic` (PMC bytes per launch) is reported only when
                profiles/roofline_traffic.json was measured on the same kernel sources (src_sha).
                MFMA-bound classes also carry `frac_of_sustained`: against the 1 930 TFLOP/s a register-operand
                MFMA stream sustains on random data at the board's power limit (profiles/round2_mfma_ceiling.txt).
  cpu_baseline  the CPU oracle (oracle/, a port of the reference path) timed on this host's
                cores on a bounded sample of the same workload.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16/f16 MFMA
MFMA_SUSTAINED_TFLOPS = 1930.0   # measured: 32x32x16 bf16, operands in registers, random data, all CUs (profiles/round2_mfma_ceiling.txt)
BATCH_PER_GPU = 256
PROF_STEPS = 3               # per-launch-event (roofline) steps, run after the timed region
IMAGE = 384
JOINTS = 11


def source_hash():
    """sha256 over the kernel sources the shipped library is built from; PMC figures measured on other sources are stale."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "spacecraft-pose-estimation_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="frames per GPU per step")
    ap.add_argument("--model", default="w48", choices=["w48", "w32"])
    ap.add_argument("--image", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--cpu-frames", type=int, default=32, help="frames of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--events", action="store_true",
                    help="BASELINE configs[4] side line: HRNet-W32 256x256, f16 MFMA kernels, mixed RGB + event-frame batch")
    ap.add_argument("--graph", type=int, default=-1,
                    help="un-profiled steps replay the forward from a hipGraph (scpose_hrnet_graph_*): 1 = every independent group of ops "
                         "on concurrent lanes (branches of a module, fuse rows, transition convolutions), 2 = only the fuse rows and "
                         "transition convolutions (short HBM-bound launches) side by side, the branches one after the other; 0: eager "
                         "launches; default: 1 (W32 batch 64: -19 %%; W48 batch 256: 27.8 vs 28.0 ms per step with 2 and 28.0 serial, "
                         "same box -- until the persistent kernels took their tiles from queues, 2 was the faster one there)")
    ap.add_argument("--fused-decode", type=int, default=1,
                    help="1: key points from scpose_hrnet_forward_decode (decode inside the network's last kernel, no heat-map round trip); "
                         "0: scpose_hrnet_forward, then scpose_decode on the side stream")
    ap.add_argument("--chained", action="store_true",
                    help="feed the PnP stage with the decoded (random-weight) keypoints instead of synthetic ones")
    ap.add_argument("--fitted", action="store_true",
                    help="side line: the small FITTED HRNet of tests/golden/chain_checkpoint.npz on synthetic landmark frames (128x128 crops "
                         "whose content determines the key points), PnP chained to the decoded key points; reports the pose error against "
                         "the generating poses.  The only configuration whose heat-maps carry a pose (VERDICT r3 #3)")
    ap.add_argument("--fitted-w48", action="store_true",
                    help="like --fitted, but HRNet-W48 384x384 (the headline geometry) on the constructed peaked-heat-map checkpoint "
                         "synthetic.w48_chain_checkpoint instead of the small fitted W16 / 128 network; reports the pose error at the chosen batch")
    ap.add_argument("--pipeline", action="store_true",
                    help="side line: files -> poses through the product CLI path (synthetic 1920x1200 JPEG frames on disk -> data loader -> "
                         "validate() -> pred.mat -> export -> opencv_poses.json), frames/s per stage beside the reference-style host loader "
                         "(tools_dev/pipeline_bench.py)")
    ap.add_argument("--pipeline-quick", action="store_true", help="--pipeline without the 4 x workers variants (their worker start-up takes minutes on a big host)")
    ap.add_argument("--pipeline-frames", type=int, default=2048)
    ap.add_argument("--pipeline-batch", type=int, default=16, help="the loader's batch (TEST.BATCH_SIZE_PER_GPU; 16 in the reference's shipped events-config.yaml:74)")
    ap.add_argument("--pipeline-workers", type=int, default=max(1, min(32, (os.cpu_count() or 1) // 4)), help="loader workers (default: what the CLI picks, parallel.auto_workers)")
    ap.add_argument("--no-chain-check", action="store_true",
                    help="skip the post-run key-point / pose check on the 64 W48 fixture frames (about 15 s: the constructed checkpoint is rebuilt from its seed)")
    ap.add_argument("--cpu-stub", action="store_true",
                    help="(tests only) run the multi-rank step loop on CPU tensors over gloo with a stand-in engine: exercises the sharding, "
                         "gather, timing and printing code, measures nothing")
    return ap.parse_args()


def cpu_baseline(cfg, sd, image, nframes, kp_sample):
    """Reference path restated on the CPU (oracle/): fp32 torch forward on all host cores,
    NumPy decode with the reference's Python loops, serial C EPnP+RANSAC per frame.
    This leg is the ONLY place bench.py touches oracle/ -- as the reported baseline, never as
    part of the measured GPU path."""
    from oracle import hrnet_ref as R, decode_ref as D, pnp_ref as P
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(123)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    bs = 16   # the reference's TEST.BATCH_SIZE_PER_GPU (landmark_regression/experiments/events/events-config.yaml)
    t_net = t_dec = t_pnp = 0.0
    done = 0
    P.lib()
    while done < nframes:
        n = min(bs, nframes - done)
        u8 = torch.randint(0, 256, (n, image, image, 3), generator=g, dtype=torch.uint8)
        t0 = time.perf_counter()
        x = (u8.permute(0, 3, 1, 2).float() / 255.0 - mean) / std
        with torch.no_grad():
            hm = R.forward(sd, cfg, x).numpy()
        t1 = time.perf_counter()
        c = np.full((n, 2), image / 2.0, dtype=np.float32)
        s = np.full((n, 2), image / 200.0 * 1.5, dtype=np.float32)
        D.decode_xyc(True, hm, c, s)
        t2 = time.perf_counter()
        P.solve_batch(kp_sample[done:done + n])
        t3 = time.perf_counter()
        t_net += t1 - t0; t_dec += t2 - t1; t_pnp += t3 - t2
        done += n
    total = t_net + t_dec + t_pnp
    return {"value": round(nframes / total, 3), "unit": "poses/s", "cores": cores, "kind": "port",
            "sample": "%d frames of the same workload (HRNet-%s %dx%d fp32 torch-CPU forward in batches of %d, NumPy decode, "
                      "serial C EPnP+RANSAC), %.1f s" % (nframes, "W48" if cfg["MODEL"]["EXTRA"]["STAGE2"]["NUM_CHANNELS"][0] == 48 else "W32",
                                                         image, image, bs, total),
            "stage_s_per_frame": {"hrnet": t_net / nframes, "decode": t_dec / nframes, "pnp": t_pnp / nframes}}


def pci_address(index):
    """'dddd:bb:dd' of visible device `index` (torch device properties), or None"""
    try:
        p = torch.cuda.get_device_properties(index)
        return "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id & 0xff, p.pci_device_id)
    except Exception:
        return None


class DeviceSampler:
    """Shader clock and board power of one GPU while the timed region runs, read from sysfs (what rocm-smi reads) by a thread of this
    process every 25 ms: lets a reader of the line tell a slow BOX (low clock / power cap) from a slow BUILD.  None when the
    files are not readable; the figures are a report, nothing is computed from them."""

    def __init__(self, index, pci=None):
        import threading
        self.sclk, self.power, self._stop, self._thread = [], [], threading.Event(), None
        # the device's own sysfs node, by PCI address (a box can hold more cards than this process may use: the n-th visible device
        # is not the n-th card in sysfs); fall back to the n-th card with a pp_dpm_sclk file
        self.dev = None
        if pci:
            for cand in (os.path.join("/sys/bus/pci/devices", pci + ".0"), os.path.join("/sys/bus/pci/devices", pci)):
                if os.path.exists(os.path.join(cand, "pp_dpm_sclk")):
                    self.dev = cand
                    break
        if self.dev is None:
            cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
            self.dev = cards[index] if index < len(cards) else None
        hw = glob.glob(os.path.join(self.dev, "hwmon", "hwmon*")) if self.dev else []
        self.pfile = next((os.path.join(h, f) for h in hw for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, f))), None)
        self._threading = threading

    def _sample(self):
        try:
            for ln in open(os.path.join(self.dev, "pp_dpm_sclk")).read().splitlines():
                if ln.rstrip().endswith("*"):
                    self.sclk.append(int("".join(ch for ch in ln.split(":")[1] if ch.isdigit())))
            if self.pfile:
                self.power.append(int(open(self.pfile).read().strip()) / 1e6)
        except (OSError, ValueError, IndexError):
            pass

    def start(self):
        if self.dev is None:
            return
        def run():
            while not self._stop.is_set():
                self._sample()
                self._stop.wait(0.025)
        self._thread = self._threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is None:
            return None
        self._stop.set(); self._thread.join()
        if not self.sclk:
            return None
        return {"sclk_mhz_median": float(np.median(self.sclk)), "sclk_mhz_min": int(min(self.sclk)), "sclk_mhz_max": int(max(self.sclk)),
                "power_w_mean": round(float(np.mean(self.power)), 1) if self.power else None,
                "power_w_max": round(float(max(self.power)), 1) if self.power else None, "samples": len(self.sclk),
                "sysfs": self.dev,
                "source": "sysfs pp_dpm_sclk / hwmon power1_average, sampled every 25 ms during the timed region"}


def chain_check(ops, syn, dev, dtype):
    """The metric's second half ("keypoint/pose err vs ref", BASELINE.json) in the driver-run line: after the timed region, the
    64 fixture frames of tests/golden/chain_w48_reference.npz go image -> key points -> pose through the SAME kernels at the headline
    geometry (HRNet-W48, 384 x 384, the constructed checkpoint synthetic.w48_chain_checkpoint: random-init weights carry no pose).
    Key points are compared with the ones the REFERENCE module + the reference's get_final_preds produced on the same weights and frames
    (stored in the fixture, made by tests/golden/make_w48_chain.py), poses with the ones the frames were rendered from.
    What this can and cannot show: DESIGN.md section 0.1 -- the checkpoint's residual and cross-branch paths are scaled by 0.02, so it
    proves the chain end to end, not that 16-bit noise never moves an arg-max of a trained network (the heat-map tests carry that)."""
    path = os.path.join(ROOT, "tests", "golden", "chain_w48_reference.npz")
    if not os.path.exists(path):
        return {"skipped": "fixture %s not found" % os.path.relpath(path, ROOT)}
    z = np.load(path)
    image, n_cand, seed, wseed = [int(v) for v in z["meta"]]
    cand = syn.landmark_frames(n_cand, np.random.default_rng(seed), image, blob_sigma=syn.W48_CHAIN_BLOB_SIGMA)
    fr = {k: v[z["test_index"]] for k, v in cand.items()}
    eng = ops.HrnetEngine(syn.w48_chain_cfg(image), syn.w48_chain_checkpoint(wseed), dtype=dtype, device=dev)
    try:
        x = torch.from_numpy(fr["crops"]).to(dev)
        c = torch.from_numpy(fr["center"]).to(dev); s = torch.from_numpy(fr["scale"]).to(dev)
        kp = eng.forward_decode(x, c, s, True)
        rows = torch.empty((kp.shape[0], 13), dtype=torch.float64, device=dev)
        ops.pnp_epnp_ransac(kp, torch.from_numpy(syn.TANGO_LANDMARKS).to(dev), torch.from_numpy(syn.SPEEDPLUS_K).to(dev),
                            torch.from_numpy(syn.SPEEDPLUS_DIST).to(dev), rows=rows)
        got, hb = kp.cpu().numpy(), rows.cpu().numpy()
    finally:
        eng.close()
    err = np.linalg.norm(got[:, :, :2] - z["ref_preds"], axis=2)
    Rg = hb[:, 0:9].reshape(-1, 3, 3)
    ang = np.arccos(np.clip((np.einsum("nij,nij->n", Rg, fr["R"]) - 1.0) / 2.0, -1.0, 1.0))
    terr = np.linalg.norm(hb[:, 9:12] - fr["t"], axis=1) / np.linalg.norm(fr["t"], axis=1)
    return {"frames": int(got.shape[0]), "joints": int(err.size), "kp_px_max": float(err.max()), "kp_px_mean": float(err.mean()),
            "maxval_abs_diff_max": float(np.abs(got[:, :, 2:3] - z["ref_maxvals"]).max()),
            "rot_err_rad_median": float(np.median(ang)), "rot_err_rad_max": float(ang.max()),
            "t_err_rel_median": float(np.median(terr)), "t_err_rel_max": float(terr.max()), "inliers_min": int(hb[:, 12].min()),
            "kp_reference": "reference pose_hrnet module (fp32) + reference get_final_preds on the same weights / frames (tests/golden/chain_w48_reference.npz)",
            "pose_reference": "the poses the frames were rendered from (landmarks drawn up to one crop pixel from their projection)",
            "workload": "HRNet-W48 384x384 constructed checkpoint (synthetic.w48_chain_checkpoint), %d fixture frames, %s, after the timed region" % (got.shape[0], dtype),
            "tolerances": "north_star: key points <= 0.5 px; rotation / translation <= 1e-4 hold against the oracle chain on identical key points (tests/test_gpu_chain.py), not against the rendered pose"}


# ---- --cpu-stub: the multi-rank step loop of THIS file on CPU tensors over gloo (tests/test_parallel_gloo.py) ----------------------
# A stand-in for torch.cuda, the engine and the two ops, so that the code a first multi-GPU run executes -- rank != 0 without host
# buffers, the double-buffered blocks, the all-gather into `gathered`, barrier + all_reduce(MAX) timing, rank 0 alone printing --
# has run before.  It computes nothing and measures nothing: the line it prints says so.
class _CpuEvent:
    def __init__(self, enable_timing=False):
        self.t = None

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _CpuStream:
    def wait_event(self, e):
        pass


class _CpuShim:
    Event = _CpuEvent

    def __init__(self):
        import contextlib
        self._null = contextlib.nullcontext

    def is_available(self): return True
    def set_device(self, d): pass
    def Stream(self, device=None): return _CpuStream()
    def current_stream(self): return _CpuStream()
    def stream(self, s): return self._null()
    def synchronize(self): pass


class _StubGraph:
    nodes = 0

    def __init__(self, eng, x, c, s):
        self.eng, self.x, self.c, self.s = eng, x, c, s

    def replay(self):
        return self.eng.forward_decode(self.x, self.c, self.s, True)


class _StubEngine:
    """forward_decode = a deterministic function of the frames (their mean), so that a wrong shard shows in the result."""
    def __init__(self, joints): self.j = joints
    def tail_fused(self, n, h, w): return True
    def forward_decode(self, x, c, s, post=True, profile=False):
        m = x.reshape(x.shape[0], -1).float().mean(1)
        return m.view(-1, 1, 1).expand(-1, self.j, 3).contiguous()
    def capture_decode(self, x, c, s, post=True, concurrent=True): return _StubGraph(self, x, c, s)
    def profile_read(self): return [{"ms": 1.0, "flops_per_frame": 1e9, "bytes_per_frame": 1e6, "kind": 1, "a": 31, "cin": 96, "cout": 96}]
    @staticmethod
    def kernel_classes(recs): return ["1:31:96:96" for _ in recs]
    def stats(self, h, w): return {"launches": 1, "flops_per_frame": 1e9, "act_bytes_per_frame": 1e6}


class _StubOps:
    """pnp_epnp_ransac(rows=...) = row i <- [rank, local frame index, mean of the frame's key points, 0..., status 11]."""
    def __init__(self, rank): self.rank = rank
    def pnp_epnp_ransac(self, kp, lm, K, dist, rows=None):
        n = kp.shape[0]
        rows.zero_()
        rows[:, 0] = float(self.rank); rows[:, 1] = torch.arange(n, dtype=torch.float64); rows[:, 2] = kp.reshape(n, -1).double().mean(1)
        rows[:, 12] = 11.0
        return rows


def main():
    args = parse()
    stub = bool(args.cpu_stub)
    cuda = _CpuShim() if stub else torch.cuda
    if args.pipeline:
        sys.path.insert(0, os.path.join(ROOT, "tools_dev"))
        import pipeline_bench
        res = pipeline_bench.run(args.pipeline_frames, args.pipeline_workers, args.pipeline_batch, args.model, quick=args.pipeline_quick)
        print(json.dumps({"metric": "frames/sec files -> poses (product CLI path)", "unit": "frames/s", **res}))
        return
    if args.events:
        args.model, args.dtype = "w32", "f16"
    if args.fitted_w48:
        args.fitted, args.model = True, "w48"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # stand-alone multi-GPU launch: this parent never touches the GPU -- the device count comes from the KFD
        # topology in sysfs, not from torch.cuda.device_count() (which can go through hipGetDeviceCount on ROCm)
        import scpose  # noqa: F401
        from importlib import import_module
        par = import_module("spacecraft-pose-estimation_amd.parallel")
        have = par.visible_gpu_count()
        if have is not None and have < args.gpus and not args.cpu_stub:
            raise SystemExit("bench.py: --gpus %d but only %d device(s) visible" % (args.gpus, have))
        raise SystemExit(par.spawn_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world), file=sys.stderr)
        args.gpus = world
    if not cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP path has no CPU fallback")
    cuda.set_device(local_rank)
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import scpose  # noqa: F401
        from importlib import import_module
        par = import_module("spacecraft-pose-estimation_amd.parallel")
        if stub:
            dist.init_process_group("gloo", init_method=par.init_method(), rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", init_method=par.init_method(), rank=rank, world_size=world, device_id=dev)

    import scpose  # noqa: F401  (alias of the hyphenated package)
    from importlib import import_module
    ops = _StubOps(rank) if stub else import_module("spacecraft-pose-estimation_amd.ops")
    syn = import_module("spacecraft-pose-estimation_amd.synthetic")   # product-side data generation (no oracle/)

    image = args.image or (IMAGE if args.model == "w48" else 256)
    chain = None
    if args.fitted_w48:
        # the headline geometry with a checkpoint whose heat-maps carry a pose: the CONSTRUCTED W48 checkpoint (synthetic.w48_chain_checkpoint)
        image = 384
        cfg = syn.w48_chain_cfg(image)
        sd = syn.w48_chain_checkpoint(0)
        chain = syn.landmark_frames(args.batch, np.random.default_rng(3000 + rank), image, blob_sigma=syn.W48_CHAIN_BLOB_SIGMA)
        args.chained = True
    elif args.fitted:
        image = 128
        cfg = syn.chain_cfg(image)
        sd = syn.load_chain_checkpoint(os.path.join(ROOT, "tests", "golden", "chain_checkpoint.npz"))
        chain = syn.landmark_frames(args.batch, np.random.default_rng(3000 + rank), image)
        args.chained = True
    else:
        cfg = syn.hrnet_cfg(48 if args.model == "w48" else 32, JOINTS, image)
        sd = None if stub else syn.random_checkpoint(cfg, seed=0)
    if stub:
        args.chained = True      # the stand-in PnP folds the decoded key points into its rows: a frame on the wrong rank shows
    eng = _StubEngine(JOINTS) if stub else ops.HrnetEngine(cfg, sd, dtype=args.dtype, device=dev)
    B = args.batch
    if args.graph < 0:
        args.graph = 1
    hh = image // 4

    # ---- synthetic inputs, resident in HBM before the timed region (shard = rank's slice) ----
    g = torch.Generator().manual_seed(1000 + rank)
    if chain is not None:
        frames = torch.from_numpy(chain["crops"]).to(dev)
        center = torch.from_numpy(chain["center"]).to(dev)
        scale = torch.from_numpy(chain["scale"]).to(dev)
    else:
        frames = (syn.mixed_batch(B, image, g) if args.events else syn.rgb_crops(B, image, g)).to(dev)
        center = torch.full((B, 2), image / 2.0, dtype=torch.float32, device=dev)
        scale = torch.full((B, 2), image / 200.0 * 1.5, dtype=torch.float32, device=dev)
    kp_np, _, _ = syn.keypoints(B, np.random.default_rng(2000 + rank), noise_px=1.0, outlier_frac=0.1)
    kp_syn = torch.from_numpy(kp_np).to(dev)
    lm = torch.from_numpy(syn.TANGO_LANDMARKS).to(dev)
    Kc = torch.from_numpy(syn.SPEEDPLUS_K).to(dev)
    dc = torch.from_numpy(syn.SPEEDPLUS_DIST).to(dev)
    # Steps are software-pipelined over two streams: decode + PnP + all-gather + D2H of step i run on a side
    # stream while the forward of step i+1 runs on the main stream (double-buffered heatmaps / result blocks /
    # pinned host buffers).  Every step's results are on the host before the closing barrier + synchronize.
    heat = [torch.empty((B, JOINTS, hh, hh), dtype=torch.float32, device=dev) for _ in range(2)]
    block = [torch.empty((B, 13), dtype=torch.float64, device=dev) for _ in range(2)]
    gathered = [torch.empty((world * B, 13), dtype=torch.float64, device=dev) for _ in range(2)] if world > 1 else None
    host_buf = [torch.empty((world * B, 13), dtype=torch.float64) if stub else torch.empty((world * B, 13), dtype=torch.float64).pin_memory()
                for _ in range(2)] if rank == 0 else None
    side = cuda.Stream(device=dev)
    # captured forward, one graph per output buffer (same kernels, same results; scpose.h: scpose_hrnet_graph_*)
    # --fused-decode 1 (default): scpose_hrnet_forward_decode -- the key points come out of the network's last kernel
    # (head_fused.hip: last fuse sum + final_layer + decode in one pass), no heat-map is written or re-read
    fused = bool(args.fused_decode) and eng.tail_fused(B, image, image)
    if args.graph:
        graphs = [eng.capture_decode(frames, center, scale, True, concurrent=args.graph) if fused
                  else eng.capture(frames, out=heat[k], concurrent=args.graph) for k in range(2)]
    else:
        graphs = None
    done = [None, None]
    counter = [0]

    prof_ms = {}
    fwd_events = []     # (start, end) events around the forward of every timed step (two records per step, no per-launch events)
    step_end_events = []   # the event that closes every timed step on the side stream: consecutive differences = per-step period

    def step(profile, timed=False):
        k = counter[0] & 1
        counter[0] += 1
        main = cuda.current_stream()
        if done[k] is not None:
            main.wait_event(done[k])              # buffers k were last read by the side stream two steps ago
        fwd_start = None
        if timed:
            fwd_start = cuda.Event(enable_timing=True)
            fwd_start.record(main)
        kp = None
        if graphs is not None and not profile:
            out = graphs[k].replay()
            kp = out if fused else None
        elif fused:
            # eager (--graph 0) and the roofline pass: the SAME launch list as the captured key-point forward (fused tail with the
            # decode inside, no heat-map written), with per-launch events when profiling
            kp = eng.forward_decode(frames, center, scale, True, profile=profile)
            if not stub:
                kp.record_stream(side)                # allocated on the main stream, read by PnP on the side stream when --chained
        else:
            eng.forward(frames, out=heat[k], profile=profile)
        fwd_done = cuda.Event(enable_timing=timed)
        fwd_done.record(main)
        if timed:
            fwd_events.append((fwd_start, fwd_done))
        with cuda.stream(side):
            side.wait_event(fwd_done)
            if kp is None:
                kp = ops.decode(heat[k], center, scale, True)
            ops.pnp_epnp_ransac(kp if args.chained else kp_syn, lm, Kc, dc, rows=block[k])   # the kernel writes [R, t, status] rows itself
            if world > 1:
                dist.all_gather_into_tensor(gathered[k], block[k])
                out = gathered[k]
            else:
                out = block[k]
            if rank == 0:
                host_buf[k].copy_(out, non_blocking=True)   # (R, t, status) of every frame on rank 0's host
            done[k] = cuda.Event(enable_timing=timed)
            done[k].record(side)
            if timed:
                step_end_events.append(done[k])
        if profile:   # per-launch HIP events of this forward (blocks the host until the forward has finished)
            recs = eng.profile_read()
            for rec, cls in zip(recs, eng.kernel_classes(recs)):
                if rec["kind"] == 9 and rec["bytes_per_frame"] == 0:
                    continue      # a convolution inside a branch chain: launched nothing, accounted under the chain's first op
                key = (rec["kind"], rec["a"], rec["cin"], rec["cout"], cls)
                e = prof_ms.setdefault(key, [0.0, 0, 0.0, 0.0])
                e[0] += rec["ms"]; e[1] += 1; e[2] += rec["flops_per_frame"] * B; e[3] += rec["bytes_per_frame"] * B
        return host_buf[k] if rank == 0 else None

    def barrier():
        if world > 1:
            dist.barrier()
        cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    sampler = DeviceSampler(local_rank, pci_address(local_rank)) if (rank == 0 and not stub) else None
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    host = None
    for i in range(args.steps):     # the timed region: exactly K steps, none of them instrumented per launch
        host = step(False, timed=True)
    barrier()
    elapsed = time.perf_counter() - t0
    device_state = sampler.stop() if sampler else None
    elapsed_own = elapsed
    rccl = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        # who took part: every rank reports its device, so that the line itself shows N ranks on N distinct devices
        props = None if stub else torch.cuda.get_device_properties(local_rank)
        me = {"rank": rank, "local_rank": local_rank, "device": "cpu (stub)" if stub else props.name,
              "pci_bus_id": None if stub else pci_address(local_rank),
              "uuid": None if stub else str(getattr(props, "uuid", "")), "ms_per_step": round(elapsed_own / args.steps * 1e3, 3), "pid": os.getpid()}
        reports = [None] * world
        dist.all_gather_object(reports, me)
        nccl_version = None
        if not stub:
            try:
                nccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                nccl_version = None
        rccl = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "nccl_version": nccl_version,
                "ranks_reporting": sorted(reports, key=lambda r: r["rank"])}
    fwd_unprofiled_ms = sum(a.elapsed_time(b) for a, b in fwd_events) / max(len(fwd_events), 1)
    # Roofline pass, AFTER the timed region: per-launch HIP events cost ~2.5 ms per forward (~280 records
    # that break back-to-back dispatch), so the steps that carry them are not part of `value`; the same pipelined step
    # (decode / PnP of the previous step beside the forward) runs PROF_STEPS more times on the same resident inputs.
    prof_steps = 0
    for i in range(PROF_STEPS):     # every rank: the step contains the all-gather
        step(True)
        prof_steps += 1
    barrier()

    if rank == 0:
        ok = int((host[:, 12] > 0).sum().item())
        total_frames = world * B * args.steps
        # ---- roofline of the dominant kernel class (largest summed time over the profiled steps) ----
        ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)     # flop per byte above which the MFMA peak binds
        total_ms = sum(v[0] for v in prof_ms.values())
        sha = source_hash()
        try:
            traffic_db = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))
        except (OSError, ValueError):
            traffic_db = {}

        def describe(key, ms, calls, flops, byts):
            """One kernel class against ITS roofline.  flops / byts: algorithmic work of the profiled launches (bytes =
            what the launch itself must move: a fused BasicBlock counts its input once + its output once)."""
            kind, a, cin, cout, cls = key
            name = {0: "stem_conv1_kernel (3->64 3x3 s2, f32 VALU)", 2: "fuse_sum_kernel (%d terms, C=%d)" % (a, cin),
                    7: "head_fused_kernel: last fuse sum (%d terms) + final_layer 1x1 %d->%d in one pass" % (a, cin, cout),
                    3: "conv_block2_kernel: fused BasicBlock 2 x (3x3 s1 %d->%d), one layer per wave, input read once + output written once" % (cin, cout),
                    4: "head_gather_kernel (k%d s%d, C=%d)" % (a // 10, a % 10, cin),
                    5: "stem_fused_kernel: conv1 + conv2 of the stem (3->64->64, both 3x3 s2), image read once + output written once",
                    6: "bottleneck_kernel: fused Bottleneck 1x1 %d->64, 3x3 64->64, 1x1 64->%d + residual, input read once + output written once" % (cin, cout),
                    8: "fuse_down_kernel: fuse row 0 + the first 3x3 s2 hop of every down path from branch 0 (%d->%d), branch 0 read once" % (cin, cout),
                    9: "conv_chain_kernel: the %d 3x3 convolutions (%d->%d) of one branch's BasicBlocks in one launch, a frame per workgroup, activations in LDS" % (a, cin, cout)}.get(
                kind, "conv %dx%d s%d %d->%d (MFMA implicit-GEMM)" % (a // 10, a // 10, a % 10, cin, cout))
            ai = flops / byts if byts else float("inf")
            sec = ms / 1e3
            if ai < ridge:
                r = {"bound": "hbm", "achieved": round(byts / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            else:
                r = {"bound": "mfma", "achieved": round(flops / sec / 1e12, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"}
                # what a register-operand MFMA stream sustains on random bf16 data at the board's power limit
                # (tools_dev/micro/mfma_bench.hip, profiles/round2_mfma_ceiling.txt); `frac` stays against the nominal peak
                r["peak_sustained_measured"] = MFMA_SUSTAINED_TFLOPS
                r["frac_of_sustained"] = round(r["achieved"] / MFMA_SUSTAINED_TFLOPS, 4)
            r["frac"] = round(r["achieved"] / r["peak"], 4)
            # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes; summary committed under
            # profiles/): only reported when it was measured on exactly these kernel sources, batch, dtype and image
            r["traffic"] = None
            tr = traffic_db.get(cls)
            if tr and tr.get("src_sha") == sha and tr.get("batch") == B and tr.get("dtype") == args.dtype and tr.get("image") == image:
                r["traffic"] = tr["fetch_bytes"] + tr["write_bytes"]
                r["traffic_unit"] = "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE), %s" % tr["kernel"]
            # The per-launch HIP-event interval contains the dispatch gap to the next launch (the profiled pass is eager, one
            # event per launch); rocprofv3's kernel durations do not.  gap = (sum of the profiled intervals of a forward - the
            # un-instrumented forward of the timed steps) / launches; the *_gap_corrected figures subtract it and are the ones
            # to compare with profiles/*_kernel_stats.csv (VERDICT r3 #6c).  `achieved` / `frac` stay the raw event figures.
            us_gc = ms / calls * 1e3 - gap_us
            if us_gc > 0:
                r["avg_launch_us_gap_corrected"] = round(us_gc, 2)
                r["achieved_gap_corrected"] = round((byts / 1e9 if r["bound"] == "hbm" else flops / 1e12) / calls / (us_gc * 1e-6), 2 if r["bound"] == "mfma" else 1)
                r["frac_gap_corrected"] = round(r["achieved_gap_corrected"] / r["peak"], 4)
            r.update({"class": cls, "kernel": name, "launches": calls, "avg_launch_us": round(ms / calls * 1e3, 2),
                      "share_of_forward": round(ms / total_ms, 4), "algorithmic_bytes_per_launch": byts / calls,
                      "algorithmic_flops_per_launch": flops / calls, "flop_per_byte": round(ai, 1),
                      "also_tflops": round(flops / sec / 1e12, 2), "also_gbs": round(byts / sec / 1e9, 1)})
            return r

        launches_per_fwd = sum(v[1] for v in prof_ms.values()) / max(prof_steps, 1)
        gap_us = max(0.0, (total_ms / max(prof_steps, 1) - fwd_unprofiled_ms) * 1e3 / max(launches_per_fwd, 1))
        ranked = sorted(prof_ms.items(), key=lambda kv: -kv[1][0])
        roof = describe(ranked[0][0], *ranked[0][1])
        roof["profiled_steps"] = prof_steps
        roof["dispatch_gap_us_per_launch"] = round(gap_us, 2)
        roof["src_sha"] = sha
        # the next kernel classes by share of the forward, each against its own roofline (same definitions);
        # class = kind:10*ksize+stride|nterms:Cin:Cout (kind 0 stem, 1 conv, 2 fuse sum, 3 fused BasicBlock, 4 head gather, 5 fused stem, 6 fused Bottleneck, 7 fused tail, 8 fuse row 0 + down hops of branch 0, 9 branch chain (all BasicBlocks of a branch); '@pixels' where a layer shape runs at two map sizes)
        roof["next_classes"] = [{k: v for k, v in describe(k2, *v2).items()
                                 if k in ("class", "share_of_forward", "avg_launch_us", "avg_launch_us_gap_corrected", "bound", "achieved", "unit", "frac", "frac_gap_corrected", "frac_of_sustained", "traffic")}
                                for k2, v2 in ranked[1:6]]
        fwd_ms = sum(v[0] for v in prof_ms.values()) / max(prof_steps, 1)   # sum of per-launch event intervals (event overhead included)
        cpu = None
        stub_check = None
        if stub:   # every rank's block must sit at its place in rank 0's host buffer, in frame order, with that rank's own frames behind it
            stub_check = True
            hb = host.numpy()
            for r in range(world):
                fr = syn.rgb_crops(B, image, torch.Generator().manual_seed(1000 + r))
                want = fr.reshape(B, -1).float().mean(1).double().numpy()
                blk = hb[r * B:(r + 1) * B]
                stub_check = stub_check and bool((blk[:, 0] == r).all() and (blk[:, 1] == np.arange(B)).all() and np.array_equal(blk[:, 2], want))
        if world == 1 and args.cpu_frames > 0 and not stub:
            cpu = cpu_baseline(cfg, sd, image, args.cpu_frames, kp_np)
        st = eng.stats(image, image)
        periods = [] if stub else [a.elapsed_time(b) for a, b in zip(step_end_events[:-1], step_end_events[1:])]
        step_ms = ({"min": round(min(periods), 3), "median": round(float(np.median(periods)), 3), "max": round(max(periods), 3), "n": len(periods),
                    "what": "time between the ends of consecutive timed steps (HIP events on the side stream, rank 0)"} if periods else None)
        if rccl is not None:      # every rank's block must be its own: key points are seeded per rank, so equal blocks mean a rank's rows were not gathered
            hb_all = host.numpy()
            blocks = [hb_all[r * B:(r + 1) * B] for r in range(world)]
            distinct = all(not np.array_equal(blocks[a], blocks[b]) for a in range(world) for b in range(a + 1, world))
            ids = [r["uuid"] or r["pci_bus_id"] for r in rccl["ranks_reporting"]]
            rccl["rank_blocks_distinct"] = bool(distinct)
            rccl["devices_distinct"] = None if stub else len(set(ids)) == world
            if not distinct or len(rccl["ranks_reporting"]) != world or [r["rank"] for r in rccl["ranks_reporting"]] != list(range(world)):
                raise SystemExit("bench.py: the gathered (R, t, status) blocks of %d ranks are not %d distinct blocks: %s" % (world, world, json.dumps(rccl)))
        chain_report = None
        if chain is not None:   # rank 0's own frames: pose of the last step against the pose each frame was rendered from
            hb = host[:B].numpy()
            Rg = hb[:, 0:9].reshape(B, 3, 3)
            cosang = np.clip((np.einsum("nij,nij->n", Rg, chain["R"]) - 1.0) / 2.0, -1.0, 1.0)
            ang = np.arccos(cosang)
            terr = np.linalg.norm(hb[:, 9:12] - chain["t"], axis=1) / np.linalg.norm(chain["t"], axis=1)
            chain_report = {"frames": B, "rot_err_rad_median": float(np.median(ang)), "rot_err_rad_max": float(ang.max()),
                            "t_err_rel_median": float(np.median(terr)), "t_err_rel_max": float(terr.max()),
                            "inliers_min": int(hb[:, 12].min()),
                            "note": "landmarks are drawn up to one crop pixel (side / 128 frame px) from their projection (synthetic.landmark_frames)"}
        # (N = 1 only: it is a property of one device's kernels, and the other ranks of a multi-GPU run should not wait ~15 s at the process
        # group's tear-down while rank 0 rebuilds the constructed checkpoint)
        if chain_report is None and world == 1 and not stub and not args.events and args.model == "w48" and args.dtype in ("bf16", "f16") and not args.no_chain_check:
            chain_report = chain_check(ops, syn, dev, args.dtype)
        line = {
            "metric": "STUB: multi-rank step loop on CPU tensors over gloo, nothing is computed or measured (--cpu-stub)" if stub else
                      "poses/sec end-to-end (HRNet+PnP) at batch 256; keypoint/pose err vs ref",
            "value": round(total_frames / elapsed, 2), "unit": "poses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%sHRNet-%s %dx%d %d joints, batch %d per GPU + batched EPnP-RANSAC HIP kernel%s%s" % (
                           "BASELINE configs[4] side line: mixed RGB + event-frame batch, " if args.events else
                           "side line: fitted chain checkpoint on synthetic landmark frames, " if args.fitted else "",
                           "W48-chain (constructed checkpoint)" if (args.fitted and image == 384) else "W16-chain" if args.fitted else args.model.upper(), image, image, JOINTS, B, " (PnP chained to decoded keypoints)" if args.chained else "",
                           "; %d frames per step frame-sharded over %d GPUs%s" % (world * B, world, " = BASELINE configs[3]" if world * B == 2048 and world == 8 else "") if world > 1 else ""),
                       "frames_per_step": world * B, "parallelism": "frame-sharded x%d, all-gather of (R,t,status); PnP/gather/D2H of step i overlap the forward of step i+1" % world,
                       "pnp_input": "decoded" if args.chained else "synthetic projected landmarks, 1 px noise, 10% outliers",
                       "forward": ("hipGraph replay, %s on concurrent lanes (%d nodes)" % ("branches, fuse rows and transitions" if args.graph == 1 else "fuse rows and transitions", graphs[0].nodes)) if graphs else "eager launches",
                       "decode": "inside the network's last kernel (scpose_hrnet_forward_decode: last fuse sum + final_layer + arg-max / quarter-pixel / back-transform, no heat-map written)" if fused
                                 else "scpose_decode on the heat-maps, on the side stream",
                       "roofline_pass": "%d eager steps with per-launch HIP events, after the timed region; same launch list as the timed forward (%s)" % (
                           prof_steps, "fused tail with the decode inside" if fused else "forward, then scpose_decode on the side stream"),
                       "decoded_keypoints": "fed to PnP" if args.chained else "computed every step and left on the device: PnP consumes the synthetic key points (SURVEY 8d: random-init heat-maps carry no pose)",
                       "launches_per_forward": st["launches"], "gflop_per_frame": round(st["flops_per_frame"] / 1e9, 3),
                       "act_mbytes_per_frame": round(st["act_bytes_per_frame"] / 1e6, 2)},
            "hrnet_forward_ms": round(fwd_unprofiled_ms, 3),          # un-instrumented forwards of the timed steps (two events per step)
            "hrnet_forward_ms_sum_of_profiled_launches": round(fwd_ms, 3),
            "hrnet_tflops": round(st["flops_per_frame"] * B / (fwd_unprofiled_ms / 1e3) / 1e12, 2),
            "poses_ok": ok, "poses_total": world * B,
            "step_ms": step_ms,
            "device_state": device_state,
            "rccl": rccl,
            "chain": chain_report,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if stub:
            line["stub_check"] = stub_check
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
