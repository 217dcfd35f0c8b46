#!/bin/bash
# round 3, second half: parity + the bench lines that the leaner laws move (run through gpurun from the repo root)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3law; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for w in cfg2 cfg3 cfg4 osc6 oscF; do
  python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline --no-strong-leg --no-streams-leg --no-extras > $O/bench_$w.json 2>$O/bench_$w.err
  python - $O/bench_$w.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(d["config"]["workload"], "step us", round(d["ms_per_step"]*1e3,3), "long", d.get("us_per_step_long_run"), "| 8M:", r["us_per_launch"], "us frac", round(r["frac"],4), r["kernel"][-40:])
PY
done
ABRK_BENCH_TS=1 python bench.py --workload osc6 --steps 500 --warmup 50 --no-cpu-baseline --no-strong-leg --no-streams-leg --no-extras > $O/bench_osc6_ts.json 2>/dev/null
python - $O/bench_osc6_ts.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("osc6_ts step us", round(d["ms_per_step"]*1e3,3), "| 8M:", r["us_per_launch"], "us frac", round(r["frac"],4))
PY
