"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
SQ_INSTS_VALU_MFMA_MOPS_BF16 pass.  usage: python tools_dev/summarize_mfma.py <pmc_dir> <out_csv>
mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE summed over the 8 XCDs): the busy counter adds up
SIMD-cycles over 256 CUs x 4 SIMDs, GRBM_GUI_ACTIVE is one cycle count per XCD."""
import collections, csv, glob, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_prof_names import short  # noqa: E402

d, out = sys.argv[1:3]
f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = short(r["Kernel_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
rows = []
for k, c in agg.items():
    n = len(calls[k])
    gui, busy, mops = c.get("GRBM_GUI_ACTIVE", 0) / n, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n, c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) / n
    if busy > 0 and gui > 0:
        rows.append((gui * n, "%s,%d,%.0f,%.0f,%.0f,%.4f" % (k, n, gui, busy, mops, busy / (128 * gui))))
rows.sort(reverse=True)
lines = ["kernel,launches,GRBM_GUI_ACTIVE_per_launch_sum_over_8_XCDs,SQ_VALU_MFMA_BUSY_CYCLES_per_launch,SQ_INSTS_VALU_MFMA_MOPS_BF16_per_launch,mfma_busy_fraction"]
lines += [r[1] for r in rows[:16]]
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
