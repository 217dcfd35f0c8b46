cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/quick5
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3
for rep in 1 2 3; do
python bench.py --workload dynF --steps 200 --warmup 20 --roofline-batch 4194304 --roofline-steps 30 --no-cpu-baseline > $O/bench_dynF_$rep.json 2> $O/err
python - <<PY
import json
d = json.load(open("$O/bench_dynF_$rep.json")); r = d["roofline"]
print("dynF cfg:", round(d["ms_per_step"]*1e3,3), "us/step | big:", r["us_per_launch"], "us", r["achieved"], "GB/s frac", r["frac"])
PY
done
