#!/usr/bin/env python3
"""Register / scratch / LDS regression check between two builds (development aid, no GPU needed):

    python tools/kernel_resources_diff.py <old build dir> <new build dir>

For every gfx950 kernel of every `*.o` present in both directories (by demangled name): prints the kernels whose
register total, scratch or LDS changed, and the ones whose waves-per-SIMD changed or that gained scratch in capitals.
Round 5 learned this the hard way: a store behind a branch in the middle of the x,y,z law cost Jaco2's kernels 14
registers - 44 B of scratch at the 256-register line and 8 % of BASELINE config 3 - and only the HBM-sized bench leg
showed it."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def table(obj):
    out = subprocess.run([sys.executable, os.path.join(HERE, "kernel_resources.py"), obj], capture_output=True, text=True).stdout
    res = {}
    for line in out.splitlines()[1:]:
        p = line.split(None, 5)
        if len(p) == 6:
            res[p[5]] = tuple(int(x) for x in p[:5])  # VGPR (total), AGPR, waves, scratch, LDS
    return res


def main():
    old, new = sys.argv[1], sys.argv[2]
    bad = 0
    for f in sorted(os.listdir(new)):
        if not f.endswith(".o") or not os.path.exists(os.path.join(old, f)):
            continue
        a, b = table(os.path.join(old, f)), table(os.path.join(new, f))
        for k in sorted(set(a) | set(b)):
            if k not in a:
                print(f"{f}: NEW      {b[k]}  {k}")
            elif k not in b:
                print(f"{f}: GONE     {a[k]}  {k}")
            elif a[k] != b[k]:
                worse = b[k][2] < a[k][2] or (b[k][3] > 0 and b[k][3] > a[k][3])
                bad += worse
                print(f"{f}: {'WORSE   ' if worse else 'changed '} {a[k]} -> {b[k]}  {k}")
    print(f"{bad} kernel(s) lost occupancy or gained scratch")


if __name__ == "__main__":
    main()
