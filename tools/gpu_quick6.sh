cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/quick6
mkdir -p $O
for g in 0 50 200 0 50 200; do
ABRK_BENCH_GRAPH=$g python bench.py --steps 2000 --warmup 200 --no-roofline-leg --no-cpu-baseline > $O/b_$g.json 2> $O/err_$g; tail -2 $O/err_$g
python - <<PY
import json
d = json.load(open("$O/b_$g.json"))
print("graph=$g cfg2:", round(d["ms_per_step"]*1e3,3), "us/step", round(d["value"]/1e6,1), "M/s")
PY
done
