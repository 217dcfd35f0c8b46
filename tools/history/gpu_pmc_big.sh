#!/bin/bash
# SQ counters of the HBM-sized leg of the given workloads: tools/gpu_pmc_big.sh cfg2 cfg4 ...
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_big; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace --output-format csv -d $O/$W -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 20 --warmup 2 --roofline-steps 5 --no-cpu-baseline --no-strong-leg --no-streams-leg > $O/$W.log 2>&1
python - "$O/$W" <<'PY'
import sys, glob, pandas as pd
f = glob.glob(sys.argv[1] + "/**/bench_counter_collection.csv", recursive=True)[0]
df = pd.read_csv(f)
df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:64]
g = df.groupby(["kernel", "Grid_Size", "Counter_Name"])["Counter_Value"].mean().unstack()
for (k, gs), r in g.iterrows():
    if gs < 1000000: continue
    w = r["SQ_WAVES"]
    print(k, gs, f"VALU/wave {r['SQ_INSTS_VALU']/w:.0f} SALU/wave {r['SQ_INSTS_SALU']/w:.0f} wave-cycles/wave {4*r['SQ_WAVE_CYCLES']/w:.0f} "
          f"valu-active {4*r['SQ_ACTIVE_INST_VALU']/w:.0f} wait-any {4*r['SQ_WAIT_ANY']/w:.0f} issue-stall {4*r['SQ_WAIT_INST_ANY']/w:.0f} busy {r['SQ_BUSY_CYCLES']:.0f}")
PY
done
