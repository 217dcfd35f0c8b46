#!/bin/bash
# executed instructions of the six-row kernels (first pass / complete program) at 8 M rows
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc6; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload osc6 --steps 50 --warmup 5 --roofline-steps 5 --sustain-seconds 0 --no-cpu-baseline --no-streams-leg --no-strong-leg --no-extras > $O/log.txt 2>&1
python - <<'PY'
import glob, os
import pandas as pd
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc6"
f = glob.glob(O + "/**/*counter_collection.csv", recursive=True)[0]
df = pd.read_csv(f)
df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str.replace("abrk::StaticArm<abrk::Tab_", "<").str[:70]
g = df.groupby(["kernel", "Grid_Size", "Counter_Name"])["Counter_Value"].mean().unstack()
for (k, gs), r in g.iterrows():
    w = r["SQ_WAVES"]
    print(f"{k:50s} grid {gs:9d} waves {w:9.0f} VALU/wave {r['SQ_INSTS_VALU']/w:9.1f} SALU/wave {r['SQ_INSTS_SALU']/w:8.1f} LDS/wave {r['SQ_INSTS_LDS']/w:7.1f} VMEM rd/wr per wave {r['SQ_INSTS_VMEM_RD']/w:7.1f}/{r['SQ_INSTS_VMEM_WR']/w:7.1f}")
PY
find $O -name "*.db" -delete
