"""One process per GPU, frames sharded by index, one all-gather of the results.

Replaces torch.nn.DataParallel(model, device_ids=cfg.GPUS) of
landmark_regression/tools/test.py:98 (single process; per-forward parameter broadcast, input
scatter, heatmap gather).  Here every rank keeps its own resident weights and owns the
contiguous slice [r*N/R, (r+1)*N/R) of the frame list (annotations[] order is preserved by
concatenation); the only exchange is an all-gather of small per-frame result rows
(RCCL over xGMI on GPUs, gloo in the CPU tests).  SURVEY.md section 8(e).
"""
import os

import torch


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    ws, rank, local = world()
    if ws == 1:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, init_method=init_method(), rank=rank, world_size=ws)
    return dist


def init_method():
    """Rendezvous of this job: the file store spawn_local_ranks() prepared (no TCP port to race for), else the
    MASTER_ADDR / MASTER_PORT environment torch.distributed.run sets."""
    f = os.environ.get("SCPOSE_RDZV_FILE")
    return "file://" + f if f else "env://"


def visible_gpu_count():
    """Number of GPUs this process would see, WITHOUT touching HIP: KFD topology nodes with SIMDs
    (/sys/class/kfd/kfd/topology/nodes/*/properties), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES.  None when the topology is not readable (the ranks then fail loudly themselves).
    A parent that is about to start one child per GPU must not initialise the runtime (torch.cuda.device_count() can go
    through hipGetDeviceCount on ROCm)."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            props = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def shard_range(n, rank, world_size):
    """Contiguous, order-preserving, balanced split of range(n): first n % R ranks get one extra."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rows(local, n_total, dist=None):
    """All-gather ragged row blocks (rank r holds rows shard_range(n_total, r, R)) into the full
    (n_total, ...) tensor on every rank, in frame order.  One collective: blocks are padded to
    the largest shard so a single all_gather_into_tensor suffices."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    ws, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, ws) for r in range(ws)]
    maxrows = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxrows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((ws * maxrows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    parts = [out[r * maxrows: r * maxrows + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, 0)


def loader_worker_context(num_workers):
    """`multiprocessing_context` for a torch DataLoader of a process that has initialised HIP: None without workers, else a
    forkserver context whose server has torch / NumPy / PIL / this package imported already.  Workers are then forked from that clean
    server -- never from this process: forking a process with a live HIP context is unsupported, and measured here a loader whose
    workers were forked from the GPU process decoded 5x slower in steady state (profiles/round5_pipeline.json) -- and start in
    tens of milliseconds each instead of re-importing torch (8 s for 8 workers, 29 s for 32 without the preload).
    CPython 3.10's fork server does not apply the parent's sys.path before it imports the preload list (and swallows the
    ImportError), so the repository root -- where the `scpose` alias module lives, which the CLIs add with sys.path.insert -- is
    put on PYTHONPATH for the server; tests/test_host.py checks that a worker starts with the package imported."""
    if not num_workers:
        return None
    import multiprocessing as mp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = os.environ.get("PYTHONPATH", "").split(os.pathsep) if os.environ.get("PYTHONPATH") else []
    if root not in paths:
        os.environ["PYTHONPATH"] = os.pathsep.join([root] + paths)
    try:
        mp.set_forkserver_preload(["torch", "numpy", "PIL.Image", "torch.utils.data", "scpose"])
    except Exception:      # the server is already running: its preload list stands
        pass
    return mp.get_context("forkserver")


def auto_workers(n_frames, cfg_workers, keep=False):
    """Loader worker processes for a data set of n_frames: cfg.WORKERS when it is set; the reference's YAML says WORKERS: 0
    (events-config.yaml:10), which decodes every frame in the CLI's own process at ~100 frames/s -- for 512 frames or more the CLIs
    use min(32, cores / 4) workers instead (same output; 8 workers 1 109, 32 workers 3 778 frames/s of decoding on a 256-core host,
    profiles/round5_pipeline.json), unless --no_auto_workers (keep=True).  Smaller sets are not worth the workers' start-up."""
    w = int(cfg_workers)
    if w > 0 or keep or n_frames < 512:
        return w
    return max(1, min(32, (os.cpu_count() or 1) // 4))


class _UnpackingLoader:
    """A DataLoader whose worker-side collate packed each batch into two tensors (dataset.collate_device_crop_packed); iterating it yields
    the usual (input, target, target_weight, meta) tuples again.  len() and the batch order are the DataLoader's."""

    def __init__(self, loader, unpack):
        self.loader, self.unpack = loader, unpack

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for packed in self.loader:
            yield self.unpack(packed)


def valid_loader(dataset, lo, hi, world_size, batch_size, workers, device_crop):
    """The DataLoader of tools/test.py:116-122 (shuffle=False, pinned) over this rank's shard [lo, hi) of `dataset`, with the build's
    extensions: workers from a clean pre-loaded fork server; with GPU crop warp, batches of packed frame windows -- and, when worker
    processes are used, every small tensor of a batch in one blob (two shared-memory segments per batch instead of fifteen)."""
    import torch.utils.data
    subset = torch.utils.data.Subset(dataset, range(lo, hi)) if world_size > 1 else dataset
    packed = bool(device_crop) and workers > 0 and hasattr(dataset, "collate_device_crop_packed")
    collate = (dataset.collate_device_crop_packed if packed else dataset.collate_device_crop) if device_crop else None
    loader = torch.utils.data.DataLoader(subset, batch_size=batch_size, shuffle=False, num_workers=workers,
                                         pin_memory=True,   # (the packed frame windows too: a background thread pins them, the copy to the device is then asynchronous)
                                         multiprocessing_context=loader_worker_context(workers), collate_fn=collate)
    return _UnpackingLoader(loader, dataset.unpack_device_crop_batch) if packed else loader


def free_port():
    """A port that was free a moment ago.  Only a default for MASTER_PORT: the ranks spawn_local_ranks() starts
    rendezvous through a file store (SCPOSE_RDZV_FILE), so nothing binds this port between the check and its use."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_local_ranks(argv, nprocs, env=None, timeout=None):
    """Start `nprocs` fresh worker processes `argv` on this node, one per rank (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment: what torch.distributed.run would set), wait for all of
    them and return the largest exit code.  Rank 0 inherits stdout (it prints the result), the others' stdout is
    discarded; stderr is inherited.  If a worker dies the rest are terminated and its code is returned.

    The caller must not have touched the GPU: workers are CHILD processes started with subprocess (never an exec of a
    process that initialised HIP), so `bench.py --gpus N` and `tools/test.py` can launch themselves without torchrun
    (replaces the in-process torch.nn.DataParallel of landmark_regression/tools/test.py:98)."""
    import subprocess
    import tempfile
    import time
    base = dict(os.environ if env is None else env)
    rdzv_dir = tempfile.mkdtemp(prefix="scpose_rdzv_")
    base.update(WORLD_SIZE=str(nprocs), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
                SCPOSE_RDZV_FILE=os.path.join(rdzv_dir, "store"))   # parallel.init_method(): file store, no port race
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    procs = []
    for r in range(nprocs):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(argv), env=e, stdout=None if r == 0 else subprocess.DEVNULL))
    t0 = time.time()
    worst = 0
    try:
        alive = set(range(nprocs))
        while alive:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0:
                    worst = rc if worst == 0 else worst
                    for q in alive:
                        procs[q].terminate()
            if timeout is not None and time.time() - t0 > timeout:
                worst = worst or 124
                for q in alive:
                    procs[q].terminate()
                timeout = None
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
        import shutil
        shutil.rmtree(rdzv_dir, ignore_errors=True)
    return worst
