"""Kernel-name shortening shared by the profile summarisers."""
import re, subprocess

def demangle(n):
    if n.startswith("_Z"):
        try:
            n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
        except Exception:
            pass
    n = re.sub(r"\(.*$", "", n)
    n = n.replace("scpose::", "").replace("void ", "")
    return n

def short(n):
    m = re.search(r"conv1x1_stream_kernelI(DF16b|DF16_)Li(\d+)ELi(\d+)ELi(\d+)E", n)     # c++filt does not know DF16b
    if m:
        return "conv1x1_stream_kernel<%s,%s,%s,%s>" % ("bf16" if m.group(1) == "DF16b" else "f16", m.group(2), m.group(3), m.group(4))
    n = demangle(n)
    m = re.search(r"(conv_\w+_kernel<[^>]*>)", n)
    if m: return m.group(1).replace(" ", "")
    m = re.search(r"conv_igemm_kernel<(.*)>", n)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        if len(a) == 5:
            return "conv_igemm<%s k%s s%s mrep%s nrep%s>" % ({"0": "bf16", "1": "f16"}.get(a[0], a[0]), a[1], a[2], a[3], a[4])
    return n[:60]

