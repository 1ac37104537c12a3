#!/usr/bin/env python3
"""profiles/rNN/pmc_valu_default_plan.json from the raw SQ / GRBM counters of the default plan's kernels (tools/prof_pmc.sh), the kernel
durations of the same GPU session (tools/prof_stats.sh) and the counters of the isolated butterfly loop (tools/microbench bfly under
rocprofv3 --pmc): which unit bounds each kernel of the headline encode.  bench.py reads the file for `roofline.bound`.

    python tools/pmc_valu.py <pmc summary.json> <kernel_stats.csv> <pmc_bfly_by_dispatch.json> <out.json>

Definitions (MI355X_MICROARCH.md, "rocprofv3 PMC slots" and "DVFS give-back"):
  cycles            = GRBM_GUI_ACTIVE / 8 XCDs                          shader cycles of one launch (the counter is summed over the XCDs)
  clock_GHz         = cycles / duration of the launch (kernel trace of the same session; the counter pass itself is not timed)
  valu_per_cycle    = SQ_INSTS_VALU / (1024 SIMDs * cycles)             VALU instructions issued per SIMD and cycle
  valu_issue_frac   = valu_per_cycle / the same ratio of the isolated radix-2 butterfly loop (gf.hpp arithmetic, 32 waves per CU, no memory):
                      1.0 = the SIMDs issue this instruction mix as fast as they can issue it at all
  hbm_frac          = algorithmic bytes / duration / 8 TB/s;  hbm_frac_achievable = the same over 6.29 TB/s (what a copy reaches)
  bound             = "valu" when valu_issue_frac >= hbm_frac_achievable, else "hbm"
SQ_ACTIVE_INST_VALU * 4 / (1024 * cycles) (the gfx94x "VALUBusy") is listed too; it exceeds 1 here because 2-cycle instructions count a
whole quad-cycle."""
import csv
import json
import re
import sys

SIMDS, XCDS = 1024, 8
HBM_PEAK, HBM_ACHIEVABLE = 8000.0, 6290.0
ALG_BYTES = 2.0 * (1 << 19) * 4096  # one pass reads and writes the 2 GiB stripe


def profile_name(template):
    m = re.search(r"ntt_tile_kernel<(\d+), (\d+), (true|false), (\d+)", template)
    if not m:
        return None
    levels, logv, w32, mode = int(m.group(1)), int(m.group(2)), m.group(3) == "true", int(m.group(4))
    return "tile_%s%d_w%s%s" % ({0: "dif", 1: "dit", 2: "mid"}.get(mode, "m%d" % mode), levels, "32" if w32 else "64", "" if logv == 5 else "_r16")


def main():
    pmc, stats, bfly, out = sys.argv[1:5]
    counters = json.load(open(pmc))
    dur = {}
    for row in csv.DictReader(open(stats)):
        name = row["Name"].replace("void ", "").replace("fastecc::", "").split("(")[0]
        dur[name] = float(row["AverageNs"]) / 1e6
    iso = json.load(open(bfly))
    loops = {}
    for k, v in iso.items():
        if "INSTS_VALU" not in " ".join(v):
            continue
        cyc = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"]) / XCDS
        if v["SQ_INSTS_VALU"][0] > 1e6:
            loops[k] = {"valu_per_cycle": v["SQ_INSTS_VALU"][0] / (SIMDS * cyc), "cycles_per_valu_instruction": SIMDS * cyc / v["SQ_INSTS_VALU"][0]}
    ref_name = "bfly_kernel<0>"  # mont: mad64 + mulhi, carry-form add and sub: the arithmetic of gf.hpp
    ref = loops[ref_name]["valu_per_cycle"]
    res = {"what": __doc__.split("\n\n")[2], "isolated_loops": loops, "yardstick": {"loop": ref_name + " (tools/microbench.hip: radix-2 butterfly, gf.hpp arithmetic)",
                                                                                   "valu_per_cycle": round(ref, 4),
                                                                                   "cycles_per_valu_instruction": round(1 / ref, 3)},
           "kernels": {}}
    for k, c in counters.items():
        pn = profile_name(k)
        if pn is None or k not in dur:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / XCDS
        vpc = c["SQ_INSTS_VALU"] / (SIMDS * cyc)
        gbps = ALG_BYTES / (dur[k] * 1e-3) / 1e9
        entry = {"template": k, "duration_ms_kernel_trace": round(dur[k], 4), "cycles": round(cyc), "clock_GHz": round(cyc / (dur[k] * 1e-3) / 1e9, 3),
                 "valu_per_cycle": round(vpc, 4), "cycles_per_valu_instruction": round(1 / vpc, 3), "valu_issue_frac": round(vpc / ref, 4),
                 "valu_busy_gfx94x_formula": round(c["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * cyc), 3),
                 "hbm_GBps_algorithmic": round(gbps, 1), "hbm_frac": round(gbps / HBM_PEAK, 4), "hbm_frac_achievable": round(gbps / HBM_ACHIEVABLE, 4),
                 "wave_cycles_split": {n: round(c[n] / c["SQ_WAVE_CYCLES"], 3) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")},
                 "raw": {n: c[n] for n in sorted(c)}}
        entry["bound"] = "valu" if entry["valu_issue_frac"] >= entry["hbm_frac_achievable"] else "hbm"
        res["kernels"][pn] = entry
    total_insts = sum(e["raw"]["SQ_INSTS_VALU"] for e in res["kernels"].values())
    total_cycles = sum(e["cycles"] for e in res["kernels"].values())
    total_ms = sum(e["duration_ms_kernel_trace"] for e in res["kernels"].values())
    floor_cycles = total_insts / SIMDS / ref
    res["encode"] = {"valu_instructions": total_insts, "cycles": total_cycles, "sum_of_kernel_ms": round(total_ms, 4),
                     "valu_floor_cycles": round(floor_cycles), "valu_floor_frac": round(floor_cycles / total_cycles, 4),
                     "valu_floor_ms_at_the_measured_clock": round(total_ms * floor_cycles / total_cycles, 4),
                     "what": "valu_floor = all VALU instructions of the three kernels issued at the isolated loop's rate, at the clock these kernels ran at"}
    json.dump(res, open(out, "w"), indent=1)
    for pn, e in res["kernels"].items():
        print(pn, e["bound"], "valu_issue_frac", e["valu_issue_frac"], "hbm_frac_achievable", e["hbm_frac_achievable"], "clock", e["clock_GHz"])
    print(res["encode"])


if __name__ == "__main__":
    main()
