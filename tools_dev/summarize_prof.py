"""Summarise rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into profiles/.
usage: python tools_dev/summarize_prof.py <stats_dir> <fetch_dir> <write_dir> <out_prefix>"""
import csv, glob, re, subprocess, sys, collections, os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_prof_names import demangle, short  # noqa: E402,F401

stats_dir, fetch_dir, write_dir, out = sys.argv[1:5]
rows = list(csv.DictReader(open((glob.glob(stats_dir + "/*/*kernel_stats.csv") + glob.glob(stats_dir + "/*kernel_stats.csv"))[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lines = ["kernel,calls,avg_us,total_ms,pct"]
for r in rows:
    if float(r["TotalDurationNs"]) / tot < 0.0005: continue
    lines.append("%s,%s,%.2f,%.3f,%.2f" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
open(out + "_kernel_stats.csv", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:16]))

def pmc(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    if not f: return {}
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != counter: continue
        k = short(r["Kernel_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    return agg
fe, wr = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
lines = ["kernel,launches,FETCH_SIZE_raw_KB_per_launch,fetch_MB_per_launch_x2_gfx950,WRITE_SIZE_KB_per_launch,write_MB_per_launch"]
for k in sorted(fe, key=lambda k: -fe[k][0]):
    f, n = fe[k]; w, nw = wr.get(k, (0, 1))
    lines.append("%s,%d,%.1f,%.2f,%.1f,%.2f" % (k, n, f / n, 2 * f / n / 1024, w / max(nw, 1), w / max(nw, 1) / 1024))
open(out + "_hbm_traffic.csv", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
