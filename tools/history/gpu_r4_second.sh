#!/bin/bash
# Round 4, second GPU call: (1) six-row law, hand-over form with mask compaction vs the round-3 scheme (same box);
# (2) workgroups of 1 / 2 / 4 wavefronts for the x,y,z kernels at the shard sizes of config 4 and at 8 M rows;
# (3) kernel trace of the 4096-row six-row step; (4) the GPU parity suite.   -> gpurun_out/r4c/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c; mkdir -p $O
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
ab() {  # file, label, workload, batch, env...
  local f=$1 lab=$2 w=$3 b=$4; shift 4
  env "$@" timeout 300 python bench.py --workload $w --batch $b $S 2> $O/err_$lab.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lab', '$w', 'B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/$f
}
: > $O/ab_osc6.txt; : > $O/ab_waves.txt
for b in 4096 16384 65536 262144; do
  ab ab_osc6.txt handover_$b osc6 $b A=1
  ab ab_osc6.txt round3_$b osc6 $b ABRK_NO_HANDOVER=1
done
ab ab_osc6.txt lane_4096 osc6 4096 ABRK_FINISH_COOP_MAX=0
ab ab_osc6.txt lane_65536 osc6 65536 ABRK_FINISH_COOP_MAX=0
ab ab_osc6.txt grid128_4096 osc6 4096 ABRK_FINISH_GRID=128
ab ab_osc6.txt j2_handover osc5_j2 4096 A=1
ab ab_osc6.txt j2_round3 osc5_j2 4096 ABRK_NO_HANDOVER=1
for w in 1 2 4; do
  for b in 4096 131072 262144 524288 1048576; do ab ab_waves.txt cfg4_w$w cfg4 $b ABRK_OSC_WAVES=$w; done
  ab ab_waves.txt cfg2_w$w cfg2 131072 ABRK_OSC_WAVES=$w
  ab ab_waves.txt cfg3_w$w cfg3 16384 ABRK_OSC_WAVES=$w
done
# HBM-sized legs, 1 / 4 wavefronts per workgroup (interleaved: same box, alternating)
for rep in 1 2; do for w in 1 4 2; do for wl in cfg2 cfg4; do
  ABRK_OSC_WAVES=$w timeout 300 python bench.py --workload $wl --steps 50 --warmup 10 --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras 2>> $O/err_big.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl waves=$w rep=$rep', r['us_per_launch'], r['frac'])" | tee -a $O/ab_waves.txt
done; done; done
# kernel trace of the 4096-row six-row step (plain launches)
cd /tmp && export TMPDIR=/tmp
ABRK_BENCH_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_osc6 -o t -- python $GRAFT_REPO_ROOT/bench.py --workload osc6 --steps 200 --warmup 20 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras > $O/trace_osc6.log 2>&1
ABRK_BENCH_GRAPH=0 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace --output-format csv -d $O/pmc_osc6 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload osc6 --steps 200 --warmup 20 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras > $O/pmc_osc6.log 2>&1
cd $GRAFT_REPO_ROOT
python - "$O" <<'PY'
import sys, glob, pandas as pd
O = sys.argv[1]
f = glob.glob(f"{O}/trace_osc6/**/t_kernel_trace.csv", recursive=True)
if f:
    df = pd.read_csv(f[0]); df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:70]
    df["us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
    print(df.groupby(["kernel", "Grid_Size_X", "Workgroup_Size_X"]).agg(n=("us", "size"), mean_us=("us", "mean"), med_us=("us", "median"), min_us=("us", "min")).to_string())
f = glob.glob(f"{O}/pmc_osc6/**/p_counter_collection.csv", recursive=True)
if f:
    df = pd.read_csv(f[0]); df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:70]
    g = df.groupby(["kernel", "Grid_Size", "Counter_Name"])["Counter_Value"].mean().unstack()
    for (k, gs), r in g.iterrows():
        w = r["SQ_WAVES"]
        print(k, gs, f"waves {w:.0f} VALU/wave {r['SQ_INSTS_VALU']/w:.0f} SALU/wave {r['SQ_INSTS_SALU']/w:.0f} wave-cycles/wave {4*r['SQ_WAVE_CYCLES']/w:.0f} valu-active {4*r['SQ_ACTIVE_INST_VALU']/w:.0f} wait-any {4*r['SQ_WAIT_ANY']/w:.0f} issue-stall {4*r['SQ_WAIT_INST_ANY']/w:.0f}")
PY
(time timeout 1500 python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
