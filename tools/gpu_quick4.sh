cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/quick4
mkdir -p $O
for spec in "ik 4096 10" "ik 262144 3" "rollout 4096 20"; do
  set -- $spec
  python bench.py --workload $1 --batch $2 --steps $3 --warmup 1 --no-roofline-leg --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; tail -2 $O/bench_$1_$2.err
  python - <<PY
import json
d = json.load(open("$O/bench_$1_$2.json"))
print("$1 B=$2:", d["ms_per_step"], "ms/launch ->", round(d["value"]/1e6,2), "M iterations/s", d["roofline_config"]["achieved"], "GB/s")
PY
done
