#!/bin/bash
# executed instructions per row of the general-chain (Jaco2) and UR5 kernels by function: tools/pmc_legs.py under rocprofv3
# --pmc.  -> gpurun_out/r6_pmc_legs/<arm>.txt
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6_pmc_legs; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for arm in jaco2 ur5; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/$arm -o p -- python $GRAFT_REPO_ROOT/tools/pmc_legs.py $arm > $O/$arm.log 2>&1
  python3 - $O/$arm $O/$arm.txt <<'PY'
import sys, glob
import pandas as pd
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)[0]
df = pd.read_csv(f)
df["kernel"] = df["Kernel_Name"].str.replace("void abrk::", "").str.split("(").str[0].str.replace("abrk::StaticArm<abrk::Tab_", "<").str.slice(0, 70)
piv = df.pivot_table(index=["kernel", "Grid_Size", "Dispatch_Id"], columns="Counter_Name", values="Counter_Value").reset_index()
g = piv.groupby(["kernel", "Grid_Size"]).mean(numeric_only=True).reset_index()
kt = pd.read_csv(glob.glob(sys.argv[1] + "/**/p_kernel_trace.csv", recursive=True)[0])
with open(sys.argv[2], "w") as fo:
    for _, r in g.iterrows():
        if r["SQ_WAVES"] < 100:
            continue
        line = f"{r['kernel']:72s} grid {int(r['Grid_Size']):9d} waves {int(r['SQ_WAVES']):7d} VALU/wave {r['SQ_INSTS_VALU'] / r['SQ_WAVES']:9.1f} SALU/wave {r['SQ_INSTS_SALU'] / r['SQ_WAVES']:8.1f}"
        print(line)
        fo.write(line + "\n")
PY
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
