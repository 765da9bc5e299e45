# kernel trace of the pipelined step: which forward kernels are stretched while decode / PnP of the previous step run
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r3trace; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
MODES=pipe rocprofv3 --kernel-trace -d $out/t -o t --output-format csv -- python3 $root/tools_dev/step_breakdown.py 256 > $out/log.txt 2>&1
cd $root
python3 - <<'PY'
import csv, os, glob, re, collections
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r3trace")
f = glob.glob(os.path.join(out, "t", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm): return re.sub(r"\(.*", "", nm.replace("void scpose::", "").replace("(anonymous namespace)::", ""))[:44]
# find the PnP kernels; print the window around the LAST one: all kernels overlapping [pnp_start - 300us, pnp_end + 300us]
pnps = [r for r in rows if "pnp_kernel" in r["Kernel_Name"]]
p = pnps[-2]
ps, pe = int(p["Start_Timestamp"]), int(p["End_Timestamp"])
with open(os.path.join(out, "window.txt"), "w") as w:
    w.write("pnp: start 0 us, end %.1f us\n" % ((pe - ps) / 1e3))
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e > ps - 600000 and s < pe + 600000:
            w.write("%9.1f %9.1f %8.1f q%s %s grid %s lds %s\n" % ((s - ps) / 1e3, (e - ps) / 1e3, (e - s) / 1e3, r["Queue_Id"], short(r["Kernel_Name"]), r["Grid_Size_X"], r["LDS_Block_Size"]))
os.remove(f)
PY
