#!/usr/bin/env python3
"""Turns the measured GF((2^61-1)^2) butterfly rates of tools/microbench_p61.hip into the time a 3-trip plan of BASELINE configs[4]
((2^20, 2^19) x 64 KB) would need, next to the 5-trip plan's measured time.

    python tools/p61_three_trip_projection.py profiles/r04/microbench_p61.jsonl [measured_5_trip_ms] [measured VALU share]

Model (DESIGN.md, 64-bit field): k = 2^19 blocks x 4096 element columns = 2^31 elements; 19 levels down + 19 up = 38 levels of 2^30
butterflies, plus one product per element for the per-block factor (counted as 2^31 products = the multiply part of two levels).
5 trips: every twiddle wave-uniform.  3 trips: tiles of 1024 blocks x 8 element columns (a wave = 8 columns x 8 blocks), four tile
halves (DIF, MID down, MID up, DIT); in each half the top three levels pair lanes (lane ^ 32, ^ 16, ^ 8) and the top two of them
also need per-lane twiddles (their exponent contains the lane's block bits), fetched by vector loads and split per lane."""
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
rate = {}
for r in rows:
    if r.get("probe") == "p61_bfly":
        rate[r["variant"]] = max(rate.get(r["variant"], 0.0), r["Gbfly_per_s"])  # best of the runs (warm clocks)
def pick(key):
    return next(v for k, v in rate.items() if key in k)
u, lane_load = pick("uniform twiddle (SGPR"), pick("loaded from a table")
x32, x16, x8 = pick("lane ^ 32"), pick("lane ^ 16"), pick("lane ^ 8")
measured5 = float(sys.argv[2]) if len(sys.argv) > 2 else 65.6
valu_share = float(sys.argv[3]) if len(sys.argv) > 3 else 58.0 / 66.0
B = 2.0**30  # butterflies per level
ms = lambda levels, g: levels * B / (g * 1e9) * 1e3
five = ms(38, u) + ms(2, u)
plain = ms(38 - 12, u)
cross = ms(4, x32) + ms(4, x16) + ms(4, x8)
lane_tw = 8 * B * (1.0 / (lane_load * 1e9) - 1.0 / (u * 1e9)) * 1e3
three = plain + cross + lane_tw + ms(2, u)
ratio = three / five
print("| quantity | value |")
print("|---|---|")
print("| butterfly, wave-uniform twiddle (what the 5-trip kernels run) | %.0f G/s |" % u)
print("| butterfly, per-lane twiddle fetched and split at each use | %.0f G/s |" % lane_load)
print("| butterfly across lane ^ 32 / ^ 16 / ^ 8 (operand in, result back) | %.0f / %.0f / %.0f G/s |" % (x32, x16, x8))
print("| 38 levels + factor, all wave-uniform (5-trip arithmetic), plain radix-2 butterflies | %.1f ms |" % five)
print("| the same work in a 3-trip plan: 26 plain levels %.1f + 12 cross-lane levels %.1f + per-lane twiddles on 8 of them %.1f + factor %.1f | %.1f ms |"
      % (plain, cross, lane_tw, ms(2, u), three))
print("| arithmetic of 3 trips / arithmetic of 5 trips | %.3f |" % ratio)
print("| 5-trip plan measured (HIP events), of which VALU-issue time (PMC, DESIGN.md) | %.1f ms, ~%.0f ms |" % (measured5, measured5 * valu_share))
print("| **3-trip plan projected: VALU-bound at** | **%.1f ms** (memory: 3 x ~12 ms, hidden) |" % (measured5 * valu_share * ratio))
print("| decision rule (VERDICT r03 item 2): build only if the projection is < 58 ms | %s |" % ("build" if measured5 * valu_share * ratio < 58 else "not built: slower than the 5-trip plan it would replace"))
