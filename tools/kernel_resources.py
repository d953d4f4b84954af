"""Compact table of per-kernel resources (VGPRs, AGPRs, spills, scratch, LDS, occupancy) from hipcc's -Rpass-analysis remarks.
Usage: python tools/kernel_resources.py [source.hip ...] [-- extra flags]   (default: every library source)"""
import re, subprocess, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import build as B
args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]
    args = args[:args.index("--")]
for src in (args or B.SOURCES):
    src = os.path.basename(src)
    cmd = [B._hipcc(), *B.COMMON_FLAGS, *B.EXTRA_FLAGS.get(src, []), *extra, "-Rpass-analysis=kernel-resource-usage", "-c",
           os.path.join(B.CSRC, src), "-o", "/dev/null"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: (\S+) \[-Rpass-analysis", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            if cur:
                print(cur)
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            cur = {"kernel": re.sub(r"\(anonymous namespace\)::", "", name)[:60]}
        elif k in ("VGPRs", "AGPRs", "VGPRs Spill", "ScratchSize", "LDS Size", "Occupancy", "TotalSGPRs"):
            cur[k] = v
    if cur:
        print(cur)
