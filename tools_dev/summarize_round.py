"""Turn the condensed profiler output of tools_dev/profile_round.sh (gpurun_out/<tag>/<workload>/summary.json, written on the
GPU box by tools_dev/condense_prof.py) into the tracked summaries under profiles/ and refresh profiles/roofline_traffic.json.

    python tools_dev/summarize_round.py <tag>

profiles/<tag>_kernel_stats.csv     headline workload, per kernel CLASS (bench.py's key kind:a:cin:cout): post-warm-up dispatches of
                                    `rocprofv3 --kernel-trace -- python3 bench.py --steps 4 --warmup 2`: median / p95 / mean / min / max,
                                    algorithmic flops and bytes per launch, TFLOP/s and GB/s at the median
profiles/<tag>_hbm_traffic.csv      per class: FETCH_SIZE x 2 (gfx950 tallies a 128-byte request as 64: MI355X_MICROARCH.md) and WRITE_SIZE
                                    per launch (separate --pmc passes) against the algorithmic bytes
profiles/<tag>_mfma_util.csv        per class: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
profiles/<tag>_lds_wait.csv         per class: LDS bank-conflict share, LDS / VMEM issue activity, wave-cycle split (wait / issue-stall / active)
profiles/<tag>_{w32_b64,events_b64}_kernel_stats.csv   BASELINE configs[1] / configs[4], per kernel symbol (captured forward: concurrent
                                    lanes reorder the dispatches, so no per-class attribution) with their PMC traffic and MFMA columns
profiles/<tag>_bench*.json          the bench lines of the same box
profiles/roofline_traffic.json      PMC bytes per launch keyed by class, stamped with the source hash / batch / dtype / image of the run"""
import json, os, shutil, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
tag = sys.argv[1]
g = os.path.join(ROOT, "gpurun_out", tag)
out = os.path.join(ROOT, "profiles", tag)


def q(s):
    return '"%s"' % s if "," in s else s


def _plain_names(summary):
    """Classes condensed with an '@pixels' suffix that is not needed (one map size per layer shape) get their plain name back."""
    names = [e["class"] for e in summary.get("class_stats", [])]
    ren = {}
    for c in names:
        base = c.split("@")[0]
        if "@" in c and sum(1 for d in names if d.split("@")[0] == base) == 1:
            ren[c] = base
    if not ren:
        return summary
    for e in summary["class_stats"]:
        e["class"] = ren.get(e["class"], e["class"])
    for key in ("work_per_launch", "pmc"):
        if key in summary:
            summary[key] = {ren.get(k, k): v for k, v in summary[key].items()}
    return summary


main = _plain_names(json.load(open(os.path.join(g, "w48_b256", "summary.json"))))
work = main["work_per_launch"]
pmc = main["pmc"]
rows = ["class,kernel,launches_per_forward,dispatches,median_us,p95_us,mean_us,min_us,max_us,share_of_forward,alg_GFLOP_per_launch,alg_MB_per_launch,TFLOPs_at_median,GBs_at_median"]
for e in main["class_stats"]:
    fl, by = work[e["class"]]
    rows.append("%s,%s,%d,%d,%.2f,%.2f,%.2f,%.2f,%.2f,%.4f,%.2f,%.2f,%.1f,%.0f" % (
        e["class"], q(e["kernel"]), e["launches_per_forward"], e["n"], e["median_us"], e["p95_us"], e["mean_us"], e["min_us"], e["max_us"],
        e["share_of_forward"], fl / 1e9, by / 1e6, fl / e["median_us"] / 1e6, by / e["median_us"] / 1e3))
for e in main["kernel_stats"]:
    if not any(e["kernel"] == c["kernel"] for c in main["class_stats"]) and e["share"] > 0.0005:
        rows.append("-,%s,,%d,%.2f,%.2f,%.2f,%.2f,%.2f,%.4f,,,," % (q(e["kernel"]), e["n"], e["median_us"], e["p95_us"], e["mean_us"], e["min_us"], e["max_us"], e["share"]))
open(out + "_kernel_stats.csv", "w").write("\n".join(rows) + "\n")

rows = ["class,kernel,fetch_MB_per_launch_x2_gfx950,write_MB_per_launch,traffic_MB_per_launch,alg_MB_per_launch,traffic_over_algorithmic"]
for e in main["class_stats"]:
    p = pmc.get(e["class"], {})
    if "FETCH_SIZE" not in p:
        continue
    fe, wr = 2 * p["FETCH_SIZE"] / 1024, p.get("WRITE_SIZE", 0) / 1024          # counters are in KB
    by = work[e["class"]][1] / 1e6
    rows.append("%s,%s,%.1f,%.1f,%.1f,%.1f,%.3f" % (e["class"], q(e["kernel"]), fe * 1.048576, wr * 1.048576, (fe + wr) * 1.048576, by, (fe + wr) * 1.048576 / by if by else 0))
open(out + "_hbm_traffic.csv", "w").write("\n".join(rows) + "\n")

# (no clock column: GRBM_GUI_ACTIVE / 8 / duration reads 2.2-4 GHz on launches this short -- MI355X_MICROARCH.md "DVFS give-back": the quotient
# is only good on dispatches of >= 10 ms; the in-kernel stamps of the development build give 1.45-1.5 GHz for these kernels)
rows = ["class,kernel,GRBM_GUI_ACTIVE_per_launch_sum_over_8_XCDs,SQ_VALU_MFMA_BUSY_CYCLES_per_launch,mfma_busy_fraction,mfma_busy_cycles_per_SIMD_over_median_us_GHz_equivalent"]
for e in main["class_stats"]:
    p = pmc.get(e["class"], {})
    if p.get("GRBM_GUI_ACTIVE", 0) > 0 and p.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
        gui = p["GRBM_GUI_ACTIVE"]
        # last column: matrix-pipe busy cycles per SIMD per microsecond of the launch = (busy fraction) x (shader clock): 1.0 would be
        # a pipe that is busy every cycle of a 1 GHz clock; with the stamps' 1.45-1.5 GHz, 0.70 means the pipe is busy 47-48 % of the cycles
        rows.append("%s,%s,%.0f,%.0f,%.4f,%.3f" % (e["class"], q(e["kernel"]), gui, p["SQ_VALU_MFMA_BUSY_CYCLES"], p["SQ_VALU_MFMA_BUSY_CYCLES"] / (128 * gui), p["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / e["median_us"] / 1e3))
open(out + "_mfma_util.csv", "w").write("\n".join(rows) + "\n")

rows = ["class,kernel,lds_bank_conflict_share_of_lds_cycles,wave_cycles_waiting,wave_cycles_issue_stalled,wave_cycles_active,issue_stall_on_lds,active_inst_lds_share,active_inst_vmem_share,mfma_valu_coexec_share_of_mfma_busy"]
for e in main["class_stats"]:
    p = pmc.get(e["class"], {})
    wc = p.get("SQ_WAVE_CYCLES", 0)
    if not wc:
        continue
    rows.append("%s,%s,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f" % (
        e["class"], q(e["kernel"]), p.get("SQ_LDS_BANK_CONFLICT", 0) / max(p.get("SQ_LDS_IDX_ACTIVE", 0), 1), p.get("SQ_WAIT_ANY", 0) / wc,
        p.get("SQ_WAIT_INST_ANY", 0) / wc, p.get("SQ_ACTIVE_INST_ANY", 0) / wc, p.get("SQ_WAIT_INST_LDS", 0) / wc,
        p.get("SQ_ACTIVE_INST_LDS", 0) / wc, p.get("SQ_ACTIVE_INST_VMEM", 0) / wc,
        p.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / max(pmc.get(e["class"], {}).get("SQ_VALU_MFMA_BUSY_CYCLES", 0), 1)))
open(out + "_lds_wait.csv", "w").write("\n".join(rows) + "\n")

for side in ("w32_b64", "events_b64"):
    f = os.path.join(g, side, "summary.json")
    if not os.path.exists(f):
        continue
    s = json.load(open(f))
    rows = ["kernel,dispatches,median_us,p95_us,mean_us,share,fetch_MB_per_launch_x2_gfx950,write_MB_per_launch,mfma_busy_fraction"]
    for e in s["kernel_stats"]:
        if e["share"] < 0.002:
            continue
        p = s["pmc"].get(e["kernel"], {})
        gui = p.get("GRBM_GUI_ACTIVE", 0)
        rows.append("%s,%d,%.2f,%.2f,%.2f,%.4f,%s,%s,%s" % (
            q(e["kernel"]), e["n"], e["median_us"], e["p95_us"], e["mean_us"], e["share"],
            "%.2f" % (2 * p["FETCH_SIZE"] / 1024 * 1.048576) if "FETCH_SIZE" in p else "",
            "%.2f" % (p["WRITE_SIZE"] / 1024 * 1.048576) if "WRITE_SIZE" in p else "",
            "%.4f" % (p["SQ_VALU_MFMA_BUSY_CYCLES"] / (128 * gui)) if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in p else ""))
    open(out + "_%s_kernel_stats.csv" % side, "w").write("\n".join(rows) + "\n")

for name in ("bench", "bench_events", "bench_w32_b64"):
    src = os.path.join(g, name + ".json")
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, out + "_%s.json" % name)
shutil.copy(os.path.join(g, "w48_b256", "bench_under_trace.json"), out + "_bench_under_trace.json")

bench = json.load(open(os.path.join(g, "w48_b256", "bench_under_trace.json")))
sha = bench["roofline"]["src_sha"]
db = {"_comment": "HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 4 --warmup 2` "
                  "(FETCH_SIZE x2: gfx950 counts a 128-byte request as 64, MI355X_MICROARCH.md), means over the post-warm-up dispatches of each "
                  "kernel class (dispatches attributed to classes by launch order, so layers that share a kernel symbol have their own "
                  "entries).  Keyed by bench.py's kernel class kind:a:cin:cout; bench.py uses an entry only when src_sha / batch / dtype / "
                  "image match its own run."}
for e in main["class_stats"]:
    p = pmc.get(e["class"], {})
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        db[e["class"]] = {"kernel": e["kernel"], "fetch_bytes": 2 * p["FETCH_SIZE"] * 1024, "write_bytes": p["WRITE_SIZE"] * 1024,
                          "batch": main["batch"], "dtype": main["dtype"], "image": 384, "src_sha": sha, "round": tag}
json.dump(db, open(os.path.join(ROOT, "profiles", "roofline_traffic.json"), "w"), indent=1)
print("profiles/%s_*: %d classes; roofline_traffic.json stamped %s" % (tag, len(main["class_stats"]), sha))
