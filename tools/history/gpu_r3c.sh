#!/bin/bash
# Round 3, third call: Mode F with C / dJ (tests + the 1128 B/row leg), the fp32 runtime-table question, and executed
# instruction counts (PMC) of the six-row first pass / cfg2 / cfg4 kernels
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3c; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -q -x -k "fused_full or return_dynamics or sharded") > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
python bench.py --workload oscFC --steps 400 --warmup 40 --roofline-batch 4194304 --no-cpu-baseline --no-strong-leg --no-streams-leg > $O/bench_oscFC.json 2> $O/bench_oscFC.err
python bench.py --workload oscF --steps 400 --warmup 40 --roofline-batch 4194304 --no-cpu-baseline --no-strong-leg --no-streams-leg > $O/bench_oscF.json 2> $O/bench_oscF.err
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3c"
for w in ("oscFC", "oscF"):
    d = json.load(open(f"{O}/bench_{w}.json")); r = d["roofline"]
    print(w, "step", d["ms_per_step"] * 1e3, "| 4M rows", r["us_per_launch"], "us", r["bytes_per_eval"], "B/row frac", r["frac"], r["kernel"][-30:])
PY
python tools/gpu_rt_fp32_check.py 8388608 > $O/rt_fp32.txt 2>&1; cat $O/rt_fp32.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --roofline-steps 5 --sustain-seconds 0 --no-cpu-baseline --no-streams-leg --no-strong-leg --no-extras --also cfg4,osc6,cfg3 > $O/pmc_sq.log 2>&1
python - <<'PY'
import glob, os
import pandas as pd
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3c"
f = glob.glob(O + "/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
df = pd.read_csv(f)
df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str.replace("abrk::StaticArm<abrk::Tab_", "<").str[:70]
g = df[df["Grid_Size"] >= 100000].groupby(["kernel", "Grid_Size", "Counter_Name"])["Counter_Value"].mean().unstack()
for (k, gs), r in g.iterrows():
    w = r["SQ_WAVES"]
    print(f"{k:60s} grid {gs:9d} waves {w:9.0f} VALU/wave {r['SQ_INSTS_VALU']/w:8.1f} SALU/wave {r['SQ_INSTS_SALU']/w:7.1f} LDS/wave {r['SQ_INSTS_LDS']/w:6.1f} VGPR {df[df['Grid_Size']==gs]['VGPR_Count'].iloc[0]}")
PY
rm -rf $O/pmc_sq/*/*.db 2>/dev/null
