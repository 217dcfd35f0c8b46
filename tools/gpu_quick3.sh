cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/quick3
mkdir -p $O
(time python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|Error" $O/pytest_gpu.log | head
python bench.py --workload rollout --steps 20 --warmup 2 --no-roofline-leg --no-cpu-baseline > $O/bench_rollout.json 2> $O/bench_rollout.err; tail -3 $O/bench_rollout.err
python - <<PY
import json
d = json.load(open("$O/bench_rollout.json"))
print("rollout B=4096 x1000 steps/launch:", d["ms_per_step"], "ms/launch ->", round(d["value"]/1e6,1), "M control steps/s (", round(d["ms_per_step"]*1e3/1000,3), "us per control step of 4096 arms)")
PY
python bench.py --workload rollout --batch 262144 --steps 5 --warmup 1 --no-roofline-leg --no-cpu-baseline > $O/bench_rollout_big.json 2> $O/bench_rollout_big.err
python - <<PY
import json
d = json.load(open("$O/bench_rollout_big.json"))
print("rollout B=262144:", d["ms_per_step"], "ms/launch ->", round(d["value"]/1e9,3), "G control steps/s")
PY
