#!/bin/bash
# Round 4: where does a 4096-row six-row step spend its time?  Kernel trace (plain launches) + SQ counters of the first
# pass and the finish kernel, hand-over form vs the round-3 inline form.   -> gpurun_out/r4b/
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S="--workload osc6 --batch ${B:-4096} --steps 200 --warmup 20 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
for mode in handover round3; do
  E="A=1"; [ $mode = round3 ] && E="ABRK_NO_HANDOVER=1"
  env $E ABRK_BENCH_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -o t -- python $GRAFT_REPO_ROOT/bench.py $S > $O/trace_$mode.log 2>&1
  env $E ABRK_BENCH_GRAPH=0 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $O/pmc_$mode -o p -- python $GRAFT_REPO_ROOT/bench.py $S > $O/pmc_$mode.log 2>&1
done
python - "$O" <<'PY'
import sys, glob, pandas as pd
O = sys.argv[1]
for mode in ("handover", "round3"):
    f = glob.glob(f"{O}/trace_{mode}/**/t_kernel_trace.csv", recursive=True)
    if f:
        df = pd.read_csv(f[0])
        df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:80]
        df["us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
        df = df.sort_values("Start_Timestamp")
        df["gap_us"] = (df["Start_Timestamp"] - df["End_Timestamp"].shift(1)) / 1e3
        g = df.groupby(["kernel", "Grid_Size", "Workgroup_Size"]).agg(n=("us", "size"), mean_us=("us", "mean"), med_us=("us", "median"), min_us=("us", "min"), gap_before_med=("gap_us", "median"))
        print(mode); print(g.to_string())
    f = glob.glob(f"{O}/pmc_{mode}/**/p_counter_collection.csv", recursive=True)
    if f:
        df = pd.read_csv(f[0])
        df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:80]
        g = df.groupby(["kernel", "Grid_Size", "Counter_Name"])["Counter_Value"].mean().unstack()
        for (k, gs), r in g.iterrows():
            w = r["SQ_WAVES"]
            print(mode, k, gs, f"waves {w:.0f} VALU/wave {r['SQ_INSTS_VALU']/w:.0f} SALU/wave {r['SQ_INSTS_SALU']/w:.0f} wave-cycles/wave {4*r['SQ_WAVE_CYCLES']/w:.0f} "
                  f"valu-active {4*r['SQ_ACTIVE_INST_VALU']/w:.0f} wait-any {4*r['SQ_WAIT_ANY']/w:.0f} issue-stall {4*r['SQ_WAIT_INST_ANY']/w:.0f} busy-cycles {r['SQ_BUSY_CYCLES']:.0f}")
PY
