#!/bin/bash
# Round 4, third GPU call: six-row hand-over form with single-wavefront finish workgroups; early-load variant of the
# first pass; kernel trace + PMC.   -> gpurun_out/r4d/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4d; mkdir -p $O
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
ab() {  # label, workload, batch, env...
  local lab=$1 w=$2 b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --batch $b $S 2> $O/err_$lab.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lab', '$w', 'B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab_osc6.txt
}
: > $O/ab_osc6.txt
V=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants/libabrk_early.so
for b in 4096 16384 65536 262144; do
  ab handover_$b osc6 $b A=1
  ab round3_$b osc6 $b ABRK_NO_HANDOVER=1
done
ab handover_32768 osc6 32768 A=1
ab round3_32768 osc6 32768 ABRK_NO_HANDOVER=1
ab handover_131072 osc6 131072 A=1
ab round3_131072 osc6 131072 ABRK_NO_HANDOVER=1
ab lane_4096 osc6 4096 ABRK_FINISH_COOP_MAX=0
ab lane_16384 osc6 16384 ABRK_FINISH_COOP_MAX=0
ab coop_16384 osc6 16384 ABRK_FINISH_COOP_MAX=100000
ab coop_65536 osc6 65536 ABRK_FINISH_COOP_MAX=100000
ab lane_65536 osc6 65536 ABRK_FINISH_COOP_MAX=0
ab j2_handover osc5_j2 4096 A=1
ab j2_round3 osc5_j2 4096 ABRK_NO_HANDOVER=1
ab j2_handover_16k osc5_j2 16384 A=1
ab j2_round3_16k osc5_j2 16384 ABRK_NO_HANDOVER=1
cd /tmp && export TMPDIR=/tmp
for B in 4096 65536; do
ABRK_BENCH_GRAPH=0 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace --output-format csv -d $O/pmc_osc6_$B -o p -- python $GRAFT_REPO_ROOT/bench.py --workload osc6 --batch $B --steps 200 --warmup 20 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras > $O/pmc_osc6_$B.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$O" <<'PY'
import sys, glob, pandas as pd
O = sys.argv[1]
for B in (4096, 65536):
    f = glob.glob(f"{O}/pmc_osc6_{B}/**/p_kernel_trace.csv", recursive=True)
    if f:
        df = pd.read_csv(f[0]); df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:70]
        df["us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
        print(B, "(durations under counter collection)"); print(df.groupby(["kernel", "Grid_Size_X", "Workgroup_Size_X"]).agg(n=("us", "size"), mean_us=("us", "mean"), med_us=("us", "median"), min_us=("us", "min")).to_string())
    f = glob.glob(f"{O}/pmc_osc6_{B}/**/p_counter_collection.csv", recursive=True)
    if f:
        df = pd.read_csv(f[0]); df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:70]
        g = df.groupby(["kernel", "Grid_Size", "Counter_Name"])["Counter_Value"].mean().unstack()
        for (k, gs), r in g.iterrows():
            w = r["SQ_WAVES"]
            print(B, k, gs, f"waves {w:.0f} VALU/wave {r['SQ_INSTS_VALU']/w:.0f} SALU/wave {r['SQ_INSTS_SALU']/w:.0f} wave-cycles/wave {4*r['SQ_WAVE_CYCLES']/w:.0f} valu-active {4*r['SQ_ACTIVE_INST_VALU']/w:.0f} wait-any {4*r['SQ_WAIT_ANY']/w:.0f} issue-stall {4*r['SQ_WAIT_INST_ANY']/w:.0f}")
PY
(time timeout 900 python -m pytest tests -m gpu -q -x -k "(six_row or osc6 or fuzz_osc or controllers_match or runtime_table or plans) and not compiled") > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
