#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table of one translation unit of fastecc_amd/csrc, from hipcc's own remarks
(-Rpass-analysis=kernel-resource-usage; cross-compiles for gfx950 without a GPU).

    python tools/resource_usage.py mixed_kernels.hip [--filter fused_radix] [--csv out.csv]
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastecc_amd import _build  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("source")
ap.add_argument("--filter", default="")
ap.add_argument("--csv", default="")
ap.add_argument("--flags", default="")
args = ap.parse_args()
src = args.source if os.path.exists(args.source) else os.path.join(_build.CSRC, args.source)
cmd = [_build.hipcc()] + _build.HIP_FLAGS + args.flags.split() + ["-I", _build.CSRC, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void |fastecc::|\(anonymous namespace\)::", "", name).split("(")[0]
        cur = {"kernel": name}
        rows.append(cur)
        continue
    m = re.search(r"remark: .*?\s+(SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
cols = ["VGPRs", "AGPRs", "SGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize", "Occupancy", "LDS Size"]
rows = [r for r in rows if args.filter in r["kernel"]]
print("%-58s %s" % ("kernel", " ".join("%11s" % c for c in cols)))
for r in rows:
    print("%-58s %s" % (r["kernel"][:58], " ".join("%11s" % r.get(c, "") for c in cols)))
if args.csv:
    with open(args.csv, "w") as f:
        f.write("kernel," + ",".join(cols) + "\n")
        for r in rows:
            f.write('"%s",' % r["kernel"] + ",".join(str(r.get(c, "")) for c in cols) + "\n")
