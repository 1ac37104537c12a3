#!/usr/bin/env python3
"""profiles/rNN/pmc_valu_p61.json: which unit bounds each tile kernel of the 64-bit field's encode at the BASELINE.json configs[4] size
(k = 2^19 blocks of 64 KB), from counters.  The same definitions as tools/pmc_valu.py; the yardstick is the isolated GF((2^61-1)^2)
butterfly loop of tools/microbench_p61.hip (bfly61_kernel<0>: twiddle limbs in SGPRs, the form the tile kernels use).

    python tools/pmc_valu_p61.py <session dir of tools/sessions/gpu_r05_p61_pmc.sh> <out.json>

The session dir holds pmc/summary.json (tools/prof_pmc_p61.sh), stats/ (rocprofv3 --kernel-trace --stats of the same command) and
pmc_loop/ (rocprofv3 --pmc of fastecc_amd/lib/microbench_p61)."""
import collections
import csv
import glob
import json
import re
import sys

SIMDS, XCDS = 1024, 8
HBM_PEAK, HBM_ACHIEVABLE = 8000.0, 6290.0
ALG_BYTES = 2.0 * (1 << 19) * 65536  # one pass reads and writes the 32 GiB stripe


def short(name):
    return name.replace("void ", "").replace("fastecc::", "").replace("p61::", "").replace("(anonymous namespace)::", "").split("(")[0]


def profile_name(template):
    """p61_tile_kernel<LOGR, LOGV, MODE, CANON, SPLIT, ...> -> the name the library's profile hooks (and bench.py) use."""
    m = re.search(r"p61_tile_kernel<(\d+), (\d+), (\d+), (true|false)", template)
    if not m:
        return None
    return "p61_tile_%s%s%s" % ({0: "dif", 1: "dit", 2: "mid"}.get(int(m.group(3)), "m" + m.group(3)), m.group(1),
                                "_canonical_output" if m.group(4) == "true" else "")


def main():
    src, out = sys.argv[1:3]
    counters = json.load(open(src + "/pmc/summary.json"))
    dur = {}
    for f in glob.glob(src + "/stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "p61_tile" in r["Name"]:
                dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e6, int(r["Calls"]))
    raw_loops = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(src + "/pmc_loop/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            raw_loops[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    loops = {}
    for k, c in raw_loops.items():
        if c.get("SQ_INSTS_VALU", [0])[0] > 1e6:
            cyc = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / XCDS
            loops[k] = {"valu_per_cycle": round(c["SQ_INSTS_VALU"][0] / (SIMDS * cyc), 4)}
    ref_name = "bfly61_kernel<0>"
    ref = loops[ref_name]["valu_per_cycle"]
    res = {"what": "VALU instructions issued per SIMD and cycle of every tile kernel of the 64-bit field's encode (k = 2^19, 64 KB blocks) against the isolated "
                   "butterfly loop's, and the algorithmic HBM rate against what a copy reaches; the larger fraction names the bound (tools/pmc_valu.py)",
           "isolated_loops": loops,
           "yardstick": {"loop": ref_name + " (tools/microbench_p61.hip: GF((2^61-1)^2) butterfly, twiddle limbs in SGPRs)", "valu_per_cycle": ref,
                         "note": "the loops differ by up to 15 % with the operand form (0.240 .. 0.281); a kernel whose mix has fewer of the 64-bit "
                                 "multiplies per instruction than the loop can read slightly above 1"},
           "kernels": {}}
    for k, c in counters.items():
        pn = profile_name(k)
        if pn is None or k not in dur:
            continue
        ms, calls = dur[k]
        cyc = c["GRBM_GUI_ACTIVE"] / XCDS
        vpc = c["SQ_INSTS_VALU"] / (SIMDS * cyc)
        gbps = ALG_BYTES / (ms * 1e-3) / 1e9
        e = {"template": k, "launches_per_encode": calls // 3, "duration_ms_kernel_trace": round(ms, 4), "cycles": round(cyc), "clock_GHz": round(cyc / ms / 1e6, 3),
             "valu_per_cycle": round(vpc, 4), "cycles_per_valu_instruction": round(1 / vpc, 3), "valu_issue_frac": round(vpc / ref, 4),
             "valu_busy_gfx94x_formula": round(c["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * cyc), 3),
             "hbm_GBps_algorithmic": round(gbps, 1), "hbm_frac": round(gbps / HBM_PEAK, 4), "hbm_frac_achievable": round(gbps / HBM_ACHIEVABLE, 4),
             "lds_bank_conflict_cycles_per_lds_active_cycle": round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_ACTIVE_INST_LDS"], 1), 4),
             "wave_cycles_split": {"SQ_WAIT_INST_ANY": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)},
             "raw": {n: c[n] for n in sorted(c)}}
        e["bound"] = "valu" if e["valu_issue_frac"] >= e["hbm_frac_achievable"] else "hbm"
        res["kernels"][pn] = e
    tot_ms = sum(e["duration_ms_kernel_trace"] * e["launches_per_encode"] for e in res["kernels"].values())
    tot_cyc = sum(e["cycles"] * e["launches_per_encode"] for e in res["kernels"].values())
    tot_inst = sum(e["raw"]["SQ_INSTS_VALU"] * e["launches_per_encode"] for e in res["kernels"].values())
    launches = sum(e["launches_per_encode"] for e in res["kernels"].values())
    res["encode"] = {"hbm_trips": launches, "sum_of_kernel_ms": round(tot_ms, 3), "cycles": tot_cyc, "clock_GHz": round(tot_cyc / tot_ms / 1e6, 3),
                     "valu_floor_frac": round(tot_inst / SIMDS / ref / tot_cyc, 4),
                     "hbm_floor_frac_achievable": round(launches * ALG_BYTES / HBM_ACHIEVABLE / 1e9 / (tot_ms * 1e-3), 4),
                     "what": "valu_floor = all VALU instructions of the encode at the isolated loop's issue rate and the measured clock; hbm_floor = the "
                             "encode's trips at a copy's rate; both as fractions of the summed kernel time"}
    json.dump(res, open(out, "w"), indent=1)
    for pn, e in res["kernels"].items():
        print(pn, e["bound"], "valu_issue_frac", e["valu_issue_frac"], "hbm_frac_achievable", e["hbm_frac_achievable"], "clock", e["clock_GHz"], "x", e["launches_per_encode"])
    print(res["encode"])


if __name__ == "__main__":
    main()
