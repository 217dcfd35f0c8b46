cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/var
mkdir -p $O
for tag in w2 w3; do
  for w in cfg2 cfg4; do
    ABRK_LIB_PATH=$GRAFT_REPO_ROOT/abr_control_amd/csrc/build/variants/libabrk_$tag.so python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_${w}_$tag.json 2> $O/bench_${w}_$tag.err
    python - <<PY
import json
d = json.load(open("$O/bench_${w}_$tag.json")); r = d["roofline"]
print("$tag $w", "cfg:", round(d["ms_per_step"]*1e3,3), "us/step | big:", r["us_per_launch"], "us", round(r["evals_per_s"]/1e9,3), "Gev/s frac", r["frac"])
PY
  done
done
