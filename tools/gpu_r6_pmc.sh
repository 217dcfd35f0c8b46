#!/bin/bash
# Executed instructions per row of the six-row first pass, by what the law is asked to do (VERDICT r5 "Next" #2a: "name
# where the 805 extra vector instructions are"): rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES on
# tools/microbench/kernel_ab.hip binaries, 1 M rows (recompute form: first pass = the PASS=1 kernel, grid 16384 blocks),
# with run-time variations of the law; differences between the lines attribute the instructions.  -> gpurun_out/<out>/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_pmc}; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # label, binary, env...
  local label=$1 bin=$2; shift; shift
  env AB_QUICK=1 "$@" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/$label -o p -- \
    $GRAFT_REPO_ROOT/tools/microbench/ab_$bin.bin $O/$label.json 1048576 > $O/$label.log 2>&1
}
for b in "$@"; do
  run ${b}_six $b A=1
  run ${b}_six_alg1 $b AB_ALG=1
  run ${b}_six_vmax $b AB_VMAX=1
  run ${b}_xyz_masked $b AB_DOF=111000
  run ${b}_abg_masked $b AB_DOF=000111
  run ${b}_five $b AB_DOF=111110
  run ${b}_nog $b AB_NOG=1
done
cd $GRAFT_REPO_ROOT
python3 - $O <<'PY'
import sys, glob, os
import pandas as pd
O = sys.argv[1]
rows = []
for d in sorted(glob.glob(f"{O}/*/")):
    f = glob.glob(d + "**/p_counter_collection.csv", recursive=True)
    if not f:
        continue
    df = pd.read_csv(f[0])
    df["kernel"] = df["Kernel_Name"].str.replace("void abrk::", "").str.split("(").str[0].str.slice(0, 90)
    piv = df.pivot_table(index=["kernel", "Grid_Size", "Dispatch_Id"], columns="Counter_Name", values="Counter_Value").reset_index()
    g = piv.groupby(["kernel", "Grid_Size"]).mean(numeric_only=True).reset_index()
    for _, r in g.iterrows():
        w = r.get("SQ_WAVES", 0)
        if w <= 0:
            continue
        rows.append((os.path.basename(d.rstrip("/")), r["kernel"], int(r["Grid_Size"]), r["SQ_INSTS_VALU"] / w, r["SQ_INSTS_SALU"] / w, int(w)))
with open(f"{O}/instr_per_wave.txt", "w") as fo:
    for r in rows:
        line = f"{r[0]:24s} {r[1]:92s} grid {r[2]:9d} waves {r[5]:7d}  VALU/wave {r[3]:9.1f}  SALU/wave {r[4]:8.1f}"
        print(line)
        fo.write(line + "\n")
PY
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*.csv" -size +2M -delete 2>/dev/null
