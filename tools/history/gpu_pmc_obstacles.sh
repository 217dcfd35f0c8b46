#!/bin/bash
# SQ counters of AvoidObstacles at 4 M rows: the redistributing kernel and the one-pass kernel  -> gpurun_out/pmc_obs/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_obs; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in lds plain; do
  E="A=1"; [ $mode = plain ] && E="ABRK_OBS_PLAIN=1"
  env $E rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $O/$mode -o p -- python $GRAFT_REPO_ROOT/bench.py --workload obstacles --steps 5 --warmup 1 --roofline-batch 4194304 --roofline-steps 5 --sustain-seconds 0 --no-cpu-baseline --no-strong-leg > $O/$mode.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$O" <<'PY'
import sys, glob, pandas as pd
O = sys.argv[1]
for mode in ("lds", "plain"):
    f = glob.glob(f"{O}/{mode}/**/p_counter_collection.csv", recursive=True)
    if not f: continue
    df = pd.read_csv(f[0]); df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:50]
    g = df.groupby(["kernel", "Grid_Size", "Counter_Name"])["Counter_Value"].mean().unstack()
    for (k, gs), r in g.iterrows():
        if gs < 100000: continue
        w = r["SQ_WAVES"]; rows = 4194304
        print(mode, k, gs, f"waves {w:.0f} VALU/row {r['SQ_INSTS_VALU']*64/rows:.0f} SALU/row {r['SQ_INSTS_SALU']*64/rows:.0f} LDS/row {r['SQ_INSTS_LDS']*64/rows:.0f} wave-cycles/row-group {4*r['SQ_WAVE_CYCLES']*64/rows:.0f} valu-active {4*r['SQ_ACTIVE_INST_VALU']*64/rows:.0f} wait-any {4*r['SQ_WAIT_ANY']*64/rows:.0f} wait-lds {4*r['SQ_WAIT_INST_LDS']*64/rows:.0f}")
PY
find $O -name "*.csv" -size +1M -delete
