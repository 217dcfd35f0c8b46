#!/bin/bash
# the bench JSON lines of profiles/round3 alone (run after tools/gpu_profiles_r3.sh + summarize_profiles.py have been
# committed, so that `roofline.traffic` of every line refers to the PMC passes of the same kernels)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3lines; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-streams-leg --no-roofline-leg > $O/bench_cfg2_k20.json 2>/dev/null
for w in cfg3 cfg4 cfg5 osc6 sliding_j2 oscF oscFC; do python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline --no-strong-leg > $O/bench_$w.json 2>/dev/null; done
ABRK_BENCH_TS=1 python bench.py --workload osc6 --steps 500 --warmup 50 --no-cpu-baseline --no-strong-leg > $O/bench_osc6_ts.json 2>/dev/null
ls -la $O | head -20
