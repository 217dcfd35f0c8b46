set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/quick
mkdir -p $O
(time python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|Error" $O/pytest_gpu.log | head
for w in cfg2 cfg3 cfg4 cfg5; do
  python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
d = json.load(open("$O/bench_$w.json"))
r = d["roofline"]
print("$w", "cfg:", round(d["ms_per_step"]*1e3,3), "us/step", round(d["value"]/1e6,1), "Mev/s | big:", r["us_per_launch"], "us", round(r["evals_per_s"]/1e9,3), "Gev/s frac", r["frac"], "| full:", d.get("roofline_full_outputs",{}).get("frac"))
PY
done
