#!/usr/bin/env python3
"""Approximate VGPR liveness of one kernel in a hipcc -S listing (straight-line approximation: branches ignored).
    python tools/isa_pressure.py /tmp/x.s <mangled-kernel-substring> [--dump N]
Prints the live-VGPR count along the kernel (every `step` lines) and the instructions around the maximum."""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
s = open(path).read()
m = re.search(r"^(\S*%s\S*):" % re.escape(key), s, re.M)
start = m.end()
end = s.find(".Lfunc_end", start)  # (a kernel may hold several s_endpgm: early exits)
body = s[start:end if end > 0 else s.index("s_endpgm", start)].splitlines()
ins = []
for l in body:
    t = l.strip()
    if not l.startswith("\t") or not t or t[0] in ".;":
        continue
    t = t.split(";")[0].strip()
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest else []
    def regs(o):
        out = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", o):
            out.update(range(int(a), int(b) + 1))
        out.update(int(x) for x in re.findall(r"\bv(\d+)\b", o))
        return out
    nodef = op.startswith(("buffer_store", "global_store", "flat_store", "ds_write", "s_", "v_cmp", "v_readlane", "v_readfirstlane")) or op.startswith("v_cmpx")
    d = set() if nodef or not ops else regs(ops[0])
    u = set()
    for o in (ops if nodef else ops[1:]):
        u |= regs(o)
    if op.startswith("v_writelane") or op.startswith("ds_write") :
        u |= d if op.startswith("v_writelane") else set()
    ins.append((t, d, u))
live = set()
counts = [0] * len(ins)
for i in range(len(ins) - 1, -1, -1):
    t, d, u = ins[i]
    live -= d
    live |= u
    counts[i] = len(live)
mx = max(counts)
at = counts.index(mx)
step = max(1, len(ins) // 60)
print("instructions", len(ins), "max live VGPRs", mx, "at", at)
print(" ".join("%d" % counts[i] for i in range(0, len(ins), step)))
n = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[3] == "--dump" else 12
for i in range(max(0, at - n), min(len(ins), at + n)):
    print("%6d %4d  %s" % (i, counts[i], ins[i][0]))
