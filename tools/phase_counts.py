#!/usr/bin/env python3
"""Instruction counts between the "; MARK <name>" comments of a kernel compiled with -DABRK_MARKS (development aid):
static counts of the ISA lines that follow each marker up to the next one, split into fp64 VALU / other VALU / LDS /
scalar.  The row programs are straight-line code, so for the main path static ~ executed (cold branches - library
sincos, velocity limiting - sit between markers too: read the numbers as upper bounds of their phase)."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else None
cur, counts, order = "start", collections.defaultdict(collections.Counter), ["start"]
for line in txt.splitlines():
    t = line.strip()
    m = re.match(r"; MARK (\S+)", t)
    if m:
        cur = m.group(1)
        if cur not in order:
            order.append(cur)
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith("v_") and "_f64" in op:
        counts[cur]["f64"] += 1
    elif op.startswith("v_"):
        counts[cur]["valu"] += 1
    elif op.startswith("ds_"):
        counts[cur]["lds"] += 1
    elif op.startswith("s_"):
        counts[cur]["salu"] += 1
    elif op.startswith(("global_", "scratch_", "buffer_")):
        counts[cur]["mem"] += 1
tot = collections.Counter()
for k in order:
    c = counts[k]
    tot.update(c)
    print(f"{k:22s} f64 {c['f64']:5d}  valu {c['valu']:5d}  lds {c['lds']:4d}  salu {c['salu']:5d}  mem {c['mem']:4d}")
print(f"{'total':22s} f64 {tot['f64']:5d}  valu {tot['valu']:5d}  lds {tot['lds']:4d}  salu {tot['salu']:5d}  mem {tot['mem']:4d}")
