#!/usr/bin/env python3
"""Print lost / repair / prepare / decode ms from the JSON line of tools/bench_decode.py (stdin)."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    for c in json.loads(line)["cases"]:
        print(c["lost"], c["repair_ms"], c["prepare_ms"], c["decode_ms"])
