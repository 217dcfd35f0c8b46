#!/bin/bash
# functional check of bench.py's contract line at N = 1 and (both ranks on device 0) N = 2
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2chk; mkdir -p $O
( time python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2>&1 | grep real
python - <<PY
import json
d = json.load(open("$O/bench_n1.json"))
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "scaling", "dtype")})
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "peak", "frac", "traffic", "us_per_launch")}, d["roofline"].get("valu"))
print("full", d["roofline_full_outputs"]["frac"], d["roofline_full_outputs"]["kernel"])
print("strong", d["strong_scaling_cfg4"])
print("cpu", {k: (v if k != "reference_cython" else (v or {}).get("evals_per_s_1core")) for k, v in d["cpu_baseline"].items() if k != "sample"})
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2000 --warmup 200 > $O/bench_n2.json 2> $O/bench_n2.err
echo "rc=$? lines=$(wc -l < $O/bench_n2.json)"; tail -3 $O/bench_n2.err
python - <<PY
import json
d = json.load(open("$O/bench_n2.json"))
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}, d["config"]["parallelism"])
print("strong", d["strong_scaling_cfg4"]); print("per gpu", d.get("roofline_per_gpu"))
PY
