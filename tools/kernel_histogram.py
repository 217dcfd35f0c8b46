#!/usr/bin/env python3
"""Static instruction histogram of one gfx950 kernel in a built object (development aid).

    python tools/kernel_histogram.py abr_control_amd/csrc/build/abrk_arm_ur5.o 'osc_kernel<abrk::StaticArm<abrk::Tab_ur5>, double, 3, false, 0>'

Prints instruction counts by class (fp64 VALU = 4 issue cycles per wave64 on gfx950's 16-lane fp64 pipe, other VALU 2,
packed fp32 counted apart), per basic block, so that straight-line main-path counts can be compared between variants
without a GPU.  The row programs are almost entirely straight-line code; loops (Jacobi sweeps) show up as blocks that
are the target of a backward branch."""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    base = os.path.join(tmp, "o.o")
    os.symlink(os.path.abspath(obj), base)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", base], cwd=tmp, check=True, capture_output=True)
    co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(tmp, co)], check=True,
                         capture_output=True, text=True).stdout
    return subprocess.run(["c++filt"], input=txt, check=True, capture_output=True, text=True).stdout


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_") and ("_f64" in op or op in ("v_fmac_f64_e32",)):
        return "valu_f64"
    if op.startswith("v_accvgpr"):
        return "valu_acc"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    txt = disassemble(obj)
    kern = None
    for m in re.finditer(r"^[0-9a-f]+ <(.*)>:$", txt, re.M):
        if pat in m.group(1):
            kern = m
            break
    if kern is None:
        sys.exit(f"no kernel matching {pat!r}")
    body = txt[kern.end():]
    nxt = re.search(r"^[0-9a-f]+ <.*>:$", body, re.M)
    body = body[: nxt.start()] if nxt else body
    tot = collections.Counter()
    ops = collections.Counter()
    for line in body.splitlines():
        line = line.strip()
        if not line or line.startswith(("//", ";")):
            continue
        op = line.split()[0]
        if op.endswith(":") or op == "s_nop" or op.startswith("s_code_end"):
            continue
        c = classify(op)
        tot[c] += 1
        ops[op] += 1
    print(kern.group(1)[:150])
    n = sum(tot.values())
    print("  total %d: " % n + ", ".join(f"{k} {v}" for k, v in sorted(tot.items())))
    cyc = 4 * tot["valu_f64"] + 2 * (tot["valu_other"] + tot["valu_pk"] + tot["valu_acc"])
    print(f"  VALU issue cycles per wave (static): {cyc}")
    if len(sys.argv) > 3:
        for op, c in ops.most_common(int(sys.argv[3])):
            print(f"    {op:28s} {c}")


if __name__ == "__main__":
    main()
