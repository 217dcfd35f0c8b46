#!/bin/bash
# kernel trace of the six-row step at 8192 / 16384 / 32768 rows (plain launches): first pass and finish kernel apart
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4pc4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 8192 16384 32768; do
ABRK_BENCH_GRAPH=0 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$b -o t -- python $GRAFT_REPO_ROOT/bench.py --workload osc6 --batch $b --steps 1000 --warmup 100 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras > $O/trace_$b.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$O" <<'PY' | tee $O/trace.txt
import sys, glob, pandas as pd
for b in (8192, 16384, 32768):
    f = glob.glob(sys.argv[1] + f"/trace_{b}/**/t_kernel_trace.csv", recursive=True)
    if f:
        d = pd.read_csv(f[0]); d["us"] = (d.End_Timestamp - d.Start_Timestamp) / 1e3
        d = d[d.Kernel_Name.str.contains("double, 6|osc6_finish")]
        d["k"] = d.Kernel_Name.str.slice(11, 40)
        g = d.groupby(["k", "Grid_Size_X", "Grid_Size_Y"]).us
        print(b); print(pd.DataFrame(dict(n=g.size(), mean_us=g.mean().round(2), med_us=g.median().round(2), min_us=g.min().round(2))).to_string())
PY
rm -rf $O/trace_*
