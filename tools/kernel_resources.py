#!/usr/bin/env python3
"""Register / LDS / scratch use of every gfx950 kernel in a built object or shared library (development aid).

    python tools/kernel_resources.py abr_control_amd/csrc/build/abrk_arm_ur5.o [substring ...]

Reads the code object's AMDGPU metadata note (what the HIP runtime itself reads).  waves/SIMD = floor(512 / (VGPRs +
AGPRs)) on gfx950's unified 512-entry register file, capped at 8."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    obj, pats = sys.argv[1], sys.argv[2:]
    tmp = tempfile.mkdtemp()
    base = os.path.join(tmp, "o.o")
    os.symlink(os.path.abspath(obj), base)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", base], cwd=tmp, check=True, capture_output=True)
    co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, co)], check=True, capture_output=True,
                         text=True).stdout
    rows = []
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda key: re.search(rf"\.{key}:\s*(\S+)", blk)
        name = g("name").group(1)
        agpr = int(blk.split("\n", 1)[0])
        vg = int(g("vgpr_count").group(1))
        rows.append((name, vg, agpr, int(g("private_segment_fixed_size").group(1)), int(g("group_segment_fixed_size").group(1)),
                     int(g("vgpr_spill_count").group(1)) if g("vgpr_spill_count") else 0))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    print(f"{'VGPR':>5} {'AGPR':>5} {'w/SIMD':>6} {'scratch':>8} {'LDS':>7}  kernel")
    for (n, vg, ag, scr, lds, sp), dn in zip(rows, names):
        dn = dn.replace("void abrk::", "").split("(")[0]
        if pats and not all(p in dn for p in pats):
            continue
        tot = vg + ag  # the metadata's vgpr_count is already the unified total on gfx90a+: see below
        waves = min(8, 512 // max(vg, 1))
        print(f"{vg:5d} {ag:5d} {waves:6d} {scr:8d} {lds:7d}  {dn[:150]}")


if __name__ == "__main__":
    main()
