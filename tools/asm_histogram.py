#!/usr/bin/env python3
"""Instruction histogram per kernel of a `hipcc --cuda-device-only -S` assembly file (development aid for counting
what a change to the row programs costs without a GPU).  Per kernel: fp64 VALU / other VALU / packed / SALU / LDS /
VMEM counts, registers, and the static VALU issue cycles (fp64 4, other 2 per wave64)."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith("v_") and "_f64" in op:
        return "f64"
    if op.startswith("v_"):
        return "v"
    if op.startswith("s_"):
        return "s"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "?"


def main():
    txt = open(sys.argv[1]).read()
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)^\s*s_endpgm", txt, re.M | re.S):
        name, body = m.group(1), m.group(2)
        tot, ops = collections.Counter(), collections.Counter()
        for line in body.splitlines():
            line = line.split(";")[0].strip()
            if not line or line.endswith(":") or line.startswith("."):
                continue
            op = line.split()[0]
            if op in ("s_nop", "s_waitcnt"):
                continue
            tot[classify(op)] += 1
            ops[op] += 1
        tail = txt[m.end(): m.end() + 3000]
        vg = re.search(r"; NumVgprs: (\d+)", tail)
        ag = re.search(r"; NumAgprs: (\d+)", tail)
        sc = re.search(r"; ScratchSize: (\d+)", tail)
        cyc = 4 * tot["f64"] + 2 * (tot["v"] + tot["pk"] + tot["acc"])
        short = re.sub(r"^_ZN?4abrk", "", name)[:60]
        print(f"{short:62s} f64 {tot['f64']:5d} v {tot['v']:4d} pk {tot['pk']:4d} acc {tot['acc']:3d} s {tot['s']:4d} lds {tot['lds']:3d} "
              f"vmem {tot['vmem']:3d} | valu {tot['f64'] + tot['v'] + tot['pk'] + tot['acc']:5d} cyc {cyc:6d} | vgpr {vg.group(1) if vg else '?'}"
              f"+{ag.group(1) if ag else '?'} scratch {sc.group(1) if sc else '?'}")
        for op, c in ops.most_common(top):
            print(f"      {op:26s} {c}")


if __name__ == "__main__":
    main()
