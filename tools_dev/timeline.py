"""Timeline statistics of the LAST forward in a rocprofv3 kernel trace: python tools_dev/timeline.py <t_kernel_trace.csv> [launches_per_forward]
prints wall time, sum of kernel durations, time with 0 / 1 / >=2 kernels in flight, and the longest kernels."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 276
rows = [r for r in rows if "scpose" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
t0 = min(int(r["Start_Timestamp"]) for r in last); t1 = max(int(r["End_Timestamp"]) for r in last)
ev = []
for r in last:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth = 0; prev = t0; hist = {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - prev); prev = t; depth += d
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
print("wall %.1f us, sum of kernels %.1f us, queues %s" % ((t1 - t0) / 1e3, tot / 1e3, sorted(set(r["Queue_Id"] for r in last))))
for k in sorted(hist): print("  %d kernels in flight: %.1f us" % (k, hist[k] / 1e3))
def short(nm): return re.sub(r"\(.*", "", nm.replace("void scpose::", "").replace("(anonymous namespace)::", ""))[:48]
by = {}
for r in last:
    k = (short(r["Kernel_Name"]), r["LDS_Block_Size"], r["Grid_Size_X"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    by.setdefault(k, []).append(d)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print("  %-50s lds %6s grid %6s  x%3d  avg %6.1f us  total %7.1f us" % (k[0], k[1], k[2], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3))
