#!/bin/bash
# SQ counters of ONE workload's config-sized launches: tools/gpu_pmc_one.sh <workload> [batch]
cd "$GRAFT_REPO_ROOT" || exit 1
W=${1:-ik}; B=${2:-0}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace --output-format csv -d $O -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload $W --batch $B --steps 5 --warmup 1 --no-roofline-leg --no-cpu-baseline > $O/log.txt 2>&1
python - "$O" <<'PY'
import sys, glob, pandas as pd
f = glob.glob(sys.argv[1] + "/**/bench_counter_collection.csv", recursive=True)[0]
df = pd.read_csv(f)
df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:70]
g = df.groupby(["kernel", "Grid_Size", "Counter_Name"])["Counter_Value"].mean().unstack()
for (k, gs), r in g.iterrows():
    w = r["SQ_WAVES"]
    print(k, gs, f"VALU/wave {r['SQ_INSTS_VALU']/w:.0f} SALU/wave {r['SQ_INSTS_SALU']/w:.0f} wave-cycles/wave {4*r['SQ_WAVE_CYCLES']/w:.0f} "
          f"valu-active {4*r['SQ_ACTIVE_INST_VALU']/w:.0f} wait-any {4*r['SQ_WAIT_ANY']/w:.0f} issue-stall {4*r['SQ_WAIT_INST_ANY']/w:.0f}")
PY
