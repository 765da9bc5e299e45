"""On the GPU box, after the rocprofv3 passes of tools_dev/profile_round.sh: condense the raw CSVs (large) into ONE json.

    python3 tools_dev/condense_prof.py <dir> <model w48|w32> <batch> <dtype> <warmup_forwards>

<dir> holds the passes that were run: trace/ (--kernel-trace), fetch/ write/ mfma/ ldsa/ ldsb/ (--pmc).  Every dispatch of a
forward kernel is attributed to bench.py's kernel CLASS (kind:a:cin:cout, the key of `roofline`) by position: the library
lists its launches in order (scpose_hrnet_profile_read), and the profiler lists the dispatches in order, so the k-th forward
kernel of a forward is the k-th op -- two layers that share a kernel symbol (192->192 and 384->384) stay apart.  Statistics
are taken over the dispatches AFTER the warm-up forwards: median, p95 and mean per class and per kernel symbol.
Passes whose dispatch count is not a whole number of forwards (hipGraph replays with concurrent lanes reorder them) are
summarised per kernel symbol only."""
import collections, csv, glob, json, os, re, statistics, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from summarize_prof_names import short  # noqa: E402

d, model, batch, dtype, warm = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5])


def op_classes():
    import torch, scpose  # noqa: F401
    from importlib import import_module
    ops = import_module("spacecraft-pose-estimation_amd.ops")
    syn = import_module("spacecraft-pose-estimation_amd.synthetic")
    image = 384 if model == "w48" else 256
    cfg = syn.hrnet_cfg(48 if model == "w48" else 32, 11, image)
    eng = ops.HrnetEngine(cfg, syn.random_checkpoint(cfg, 0), dtype=dtype)
    x = torch.randint(0, 256, (batch, image, image, 3), dtype=torch.uint8, device="cuda")
    eng.forward(x, profile=True)
    recs = [r for r in eng.profile_read() if not (r["kind"] in (2, 9) and r["bytes_per_frame"] == 0)]   # the fuse row the fused tail absorbs and the later convolutions of a branch chain launch nothing
    names = eng.kernel_classes(recs)
    # algorithmic work per launch of a class = the MEAN over its launches (a class mixes launches with and without a residual:
    # 96 -> 96 moves 339.7 MB with one and 226.5 MB without; bench.py's roofline line divides by the same mean -- VERDICT r4)
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for c, r in zip(names, recs):
        a = acc[c]; a[0] += r["flops_per_frame"] * batch; a[1] += r["bytes_per_frame"] * batch; a[2] += 1
    return names, {c: (a[0] / a[2], a[1] / a[2]) for c, a in acc.items()}


def find(sub, pat):
    f = glob.glob(os.path.join(d, sub, "**", pat), recursive=True)
    return f[0] if f else None


def is_forward(name):
    return ("scpose" in name) and not any(t in name for t in ("decode_kernel", "pnp_kernel", "crop_", "flip_merge", "heatmap_acc", "max_preds"))


classes, work = op_classes()
nops = len(classes)
out = {"model": model, "batch": batch, "dtype": dtype, "launches_per_forward": nops, "warmup_forwards_dropped": warm, "work_per_launch": work}


def pct(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(round(p * (len(v) - 1))))]


def stats(v):
    return {"n": len(v), "median_us": statistics.median(v), "p95_us": pct(v, 0.95), "mean_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}


tr = find("trace", "*kernel_trace.csv")
if tr:
    rows = list(csv.DictReader(open(tr)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    fw = [r for r in rows if is_forward(r["Kernel_Name"])]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by_name = collections.defaultdict(list)
    nfwd = len(fw) // nops if len(fw) % nops == 0 else 0
    out["forwards_traced"] = nfwd
    by_class = collections.defaultdict(list); kern_of = {}
    for i, r in enumerate(fw):
        if nfwd and i // nops < warm:
            continue
        by_name[short(r["Kernel_Name"])].append(dur(r))
        if nfwd:
            by_class[classes[i % nops]].append(dur(r)); kern_of[classes[i % nops]] = short(r["Kernel_Name"])
    for r in rows:
        if not is_forward(r["Kernel_Name"]):
            by_name[short(r["Kernel_Name"])].append(dur(r))
    tot = sum(sum(v) for k, v in by_name.items())
    out["kernel_stats"] = sorted(({"kernel": k, "share": sum(v) / tot, **stats(v)} for k, v in by_name.items()), key=lambda e: -e["share"])
    if nfwd:
        ftot = sum(sum(v) for v in by_class.values())
        out["class_stats"] = sorted(({"class": c, "kernel": kern_of[c], "launches_per_forward": classes.count(c), "share_of_forward": sum(v) / ftot, **stats(v)}
                                     for c, v in by_class.items()), key=lambda e: -e["share_of_forward"])
        out["forward_sum_of_kernels_ms"] = ftot / max(nfwd - warm, 1) / 1e3
    os.remove(tr)

pmc = {}
for sub in ("fetch", "write", "mfma", "ldsa", "ldsb"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))    # dispatch -> counter -> value
    name = {}
    for r in csv.DictReader(open(f)):
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    ids = [i for i in sorted(per) if is_forward(name[i])]
    whole = len(ids) % nops == 0
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, i in enumerate(ids):
        if whole and k // nops < warm:
            continue
        key = classes[k % nops] if whole else short(name[i])
        for c, v in per[i].items():
            acc[key][c].append(v)
    for i in sorted(per):                                                        # the tail kernels (decode, PnP), by symbol
        if not is_forward(name[i]):
            for c, v in per[i].items():
                acc[short(name[i])][c].append(v)
    for key, cs in acc.items():
        for c, v in cs.items():
            pmc.setdefault(key, {})[c] = sum(v) / len(v)
            pmc[key]["_launches_" + sub] = len(v)
    out.setdefault("pmc_keyed_by", {})[sub] = "class" if whole else "kernel symbol"
    os.remove(f)
out["pmc"] = pmc
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print("condensed:", d, "forwards traced", out.get("forwards_traced"), "classes", len(out.get("class_stats", [])), "pmc keys", len(pmc))
