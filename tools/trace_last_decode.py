#!/usr/bin/env python3
"""Kernel durations of the last fastecc_decode in a rocprofv3 kernel trace CSV: python tools/trace_last_decode.py <kernel_trace.csv> [count]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ours = [r for r in rows if "fastecc" in r["Kernel_Name"]]
for r in ours[-count:]:
    name = r["Kernel_Name"].replace("fastecc::", "").replace("(anonymous namespace)::", "")
    print("%-90s %9.1f us" % (name[:90], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
