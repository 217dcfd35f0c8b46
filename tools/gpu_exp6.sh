set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp6
(time python -m pytest tests -m gpu -q) > gpurun_out/exp6/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/exp6/pytest_gpu.log
for w in cfg2 dynF; do
  python bench.py --workload $w --steps 500 --warmup 50 --roofline-batch 4194304 --no-cpu-baseline > gpurun_out/exp6/bench_$w.json 2> gpurun_out/exp6/bench_$w.err
  tail -3 gpurun_out/exp6/bench_$w.err
  python - <<PY
import json
d = json.load(open("gpurun_out/exp6/bench_$w.json"))
print("$w", "cfg:", d["ms_per_step"]*1e3, "us/step", d["value"]/1e6, "Mev/s | big:", d["roofline"]["us_per_launch"], "us", d["roofline"]["evals_per_s"]/1e9, "Gev/s", d["roofline"]["achieved"], "GB/s frac", d["roofline"]["frac"])
PY
done
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exp6/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --workload dynF --steps 20 --warmup 5 --roofline-steps 3 --roofline-batch 4194304 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/exp6/pmc_$c.log 2>&1
done
