#!/usr/bin/env python3
"""Skeleton of the biggest loop of /tmp/isa/cfg3.s: memory instructions, waits, barriers, branches,
with the number of VALU / SALU instructions between them (tuning aid)."""
import re, sys
lines = [l.rstrip() for l in open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa/cfg3.s")]
# loop = from the label that the last backward branch targets to that branch
labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
best = (0, 0, 0)
for i, l in enumerate(lines):
    m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
        best = (i - labels[m.group(1)], labels[m.group(1)], i)
_, a, b = best
nv = ns = 0
tot = {}
for l in lines[a:b + 1]:
    t = l.strip()
    if not t or t.startswith(";"): continue
    op = t.split()[0]
    if op.startswith("."):
        print("  [%d valu %d salu]" % (nv, ns)); nv = ns = 0
        print(t[:60]); continue
    tot[op.split("_")[0]] = tot.get(op.split("_")[0], 0) + 1
    if op.startswith(("ds_", "global_", "s_waitcnt", "s_barrier", "s_cbranch", "s_branch", "s_load", "buffer_", "flat_", "scratch_")):
        if nv or ns: print("  [%d valu %d salu]" % (nv, ns)); nv = ns = 0
        print("    " + t[:100])
    elif op.startswith("v_"): nv += 1
    elif op.startswith("s_"): ns += 1
print(tot, "lines", b - a)
