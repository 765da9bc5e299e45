"""Condense the raw rocprofv3 output of tools_dev/profile_round.sh into profiles/<tag>_*.csv / .json and refresh
profiles/roofline_traffic.json (PMC bytes per launch of the dominant kernel classes, stamped with the source hash,
batch, dtype and image size of the bench run they were measured on -- bench.py reports `roofline.traffic` only when all
of them match).   usage: python tools_dev/summarize_round.py <tag>"""
import csv, glob, json, os, shutil, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from summarize_prof_names import short  # noqa: E402

tag = sys.argv[1]
g = os.path.join(ROOT, "gpurun_out")
out = os.path.join(ROOT, "profiles", tag)
subprocess.check_call([sys.executable, os.path.join(HERE, "summarize_prof.py"), "%s/%s_stats" % (g, tag), "%s/%s_fetch" % (g, tag),
                       "%s/%s_write" % (g, tag), out])
subprocess.check_call([sys.executable, os.path.join(HERE, "summarize_mfma.py"), "%s/%s_mfma" % (g, tag), out + "_mfma_util.csv"])
for name in ("bench", "bench_under_prof", "bench_events", "bench_w32_b64"):
    src = "%s/%s_%s.json" % (g, tag, name)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, out + "_%s.json" % name)
bench = json.load(open("%s/%s_bench_under_prof.json" % (g, tag)))
sha = bench["roofline"]["src_sha"]
traffic = {}
for line in open(out + "_hbm_traffic.csv").read().splitlines()[1:]:       # kernel names contain commas: split from the right
    name, launches, fraw, fmb, wraw, wmb = line.rsplit(",", 5)
    traffic[name] = {"fetch_MB_per_launch_x2_gfx950": fmb, "write_MB_per_launch": wmb}
# bench.py kernel class -> profiler kernel symbol (bf16, W48 384x384 batch 256)
CLASS_KERNEL = {"3:31:48:48": "conv_block_kernel<0,3>", "1:31:96:96": "conv_m32p_kernel<0,3,1,3,3,6>",
                "1:31:192:192": "conv_m32p_kernel<0,3,1,3,3,0>", "1:31:384:384": "conv_m32p_kernel<0,3,1,3,3,0>"}
db = {"_comment": "HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 4 --warmup 2` "
                  "(FETCH_SIZE x2: gfx950 counts a 128-byte request as 64, MI355X_MICROARCH.md).  Keyed by bench.py's kernel class "
                  "kind:a:cin:cout; bench.py uses an entry only when src_sha / batch / dtype / image match its own run.  The 192->192 and "
                  "384->384 classes share one kernel symbol: their entry is the average over both."}
for cls, kern in CLASS_KERNEL.items():
    r = traffic.get(kern)
    if not r:
        continue
    db[cls] = {"kernel": kern + (" (192->192 and 384->384 launches averaged)" if cls.endswith(("192", "384")) else ""),
               "fetch_bytes": float(r["fetch_MB_per_launch_x2_gfx950"]) * 1024 * 1024, "write_bytes": float(r["write_MB_per_launch"]) * 1024 * 1024,
               "batch": 256, "dtype": "bf16", "image": 384, "src_sha": sha, "round": tag}
json.dump(db, open(os.path.join(ROOT, "profiles", "roofline_traffic.json"), "w"), indent=1)
print("roofline_traffic.json:", {k: v["src_sha"] for k, v in db.items() if k != "_comment"})
