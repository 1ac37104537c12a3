#!/usr/bin/env python3
"""Write profiles/rNN/counters_stamp.json: which sources and which kernels the committed counter files of that directory describe.

    python tools/stamp_counters.py profiles/r06 [session name]

bench.py quotes roofline.bound / traffic / frac_rocprof from those files (they are not measured in the driver's run).  The stamp ties them
to the binary: the sha256 of the headline kernels' sources (fastecc_amd/_build.kernel_sources_sha256) and the kernel templates / profile
names the files hold.  When the tree's hash or the loaded library's kernels differ, bench.py labels the quoted fields STALE instead."""
import csv
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastecc_amd import _build  # noqa: E402


def main():
    d = sys.argv[1]
    session = sys.argv[2] if len(sys.argv) > 2 else ""
    stamp = {"what": __doc__.split("\n\n")[2], "made": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()), "session": session,
             "sources": _build.kernel_sources_sha256(), "hip_flags": _build.HIP_FLAGS, "files": {}, "kernel_templates": [], "profile_names": []}
    stats = os.path.join(d, "rocprofv3_kernel_stats_bench_default.csv")
    if os.path.exists(stats):
        for row in csv.DictReader(open(stats)):
            m = re.search(r"ntt_tile_kernel<[^>]*>", row.get("Name", ""))
            if m:
                stamp["kernel_templates"].append(m.group(0))
        stamp["files"]["rocprofv3_kernel_stats_bench_default.csv"] = "rocprofv3 --kernel-trace --stats of the default bench.py run"
    valu = os.path.join(d, "pmc_valu_default_plan.json")
    if os.path.exists(valu):
        stamp["profile_names"] = sorted(json.load(open(valu)).get("kernels", {}))
        stamp["files"]["pmc_valu_default_plan.json"] = "tools/pmc_valu.py"
    if os.path.exists(os.path.join(d, "pmc_traffic.json")):
        stamp["files"]["pmc_traffic.json"] = "tools/prof_pmc.sh: FETCH_SIZE / WRITE_SIZE passes"
    json.dump(stamp, open(os.path.join(d, "counters_stamp.json"), "w"), indent=1)
    print(json.dumps(stamp)[:400])


if __name__ == "__main__":
    main()
