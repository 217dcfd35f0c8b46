#!/bin/bash
# timing experiments on the six-row first pass (variants with WRONG results on purpose): what do the record stores and the
# deferral branch cost a 4096-row step?  kernel trace, plain launches   -> gpurun_out/r4x/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in base norec nodefer; do
  L=$GRAFT_REPO_ROOT/abr_control_amd/libabrk.so; [ $v != base ] && L=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants/libabrk_$v.so
  ABRK_LIB_PATH=$L ABRK_BENCH_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$v -o t -- python $GRAFT_REPO_ROOT/bench.py --workload osc6 --steps 200 --warmup 20 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras > $O/t_$v.log 2>&1
  ABRK_LIB_PATH=$L ABRK_BENCH_GRAPH=0 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $O/p_$v -o p -- python $GRAFT_REPO_ROOT/bench.py --workload osc6 --steps 200 --warmup 20 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras > $O/p_$v.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$O" <<'PY'
import sys, glob, pandas as pd
O = sys.argv[1]
for v in ("base", "norec", "nodefer"):
    f = glob.glob(f"{O}/t_{v}/**/t_kernel_trace.csv", recursive=True)
    if f:
        df = pd.read_csv(f[0]); df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:60]
        df["us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
        g = df.groupby(["kernel"]).agg(n=("us", "size"), mean_us=("us", "mean"), min_us=("us", "min"))
        print(v); print(g.to_string())
    f = glob.glob(f"{O}/p_{v}/**/p_counter_collection.csv", recursive=True)
    if f:
        df = pd.read_csv(f[0]); df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:60]
        g = df.groupby(["kernel", "Counter_Name"])["Counter_Value"].mean().unstack()
        for k, r in g.iterrows():
            w = r["SQ_WAVES"]
            print(v, k, f"waves {w:.0f} VALU {r['SQ_INSTS_VALU']/w:.0f} SALU {r['SQ_INSTS_SALU']/w:.0f} LDS {r['SQ_INSTS_LDS']/w:.0f} wave-cycles {4*r['SQ_WAVE_CYCLES']/w:.0f} valu-active {4*r['SQ_ACTIVE_INST_VALU']/w:.0f} wait-any {4*r['SQ_WAIT_ANY']/w:.0f} wait-lds {4*r['SQ_WAIT_INST_LDS']/w:.0f}")
PY
find $O -name "*.csv" -size +1M -delete
