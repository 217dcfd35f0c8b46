#!/bin/bash
# quick A/B on the GPU box: parity tests + the HBM-sized roofline legs of the BASELINE configs
# usage (through gpurun): bash tools/gpu_r2_quick.sh <tag> [workloads...]
tag=${1:-quick}; shift
wl=${@:-cfg2 cfg3 cfg4 cfg5}
out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for w in $wl; do
  python bench.py --workload $w --no-cpu-baseline --no-streams-leg --steps 400 --warmup 100 > $out/bench_$w.json 2> $out/bench_$w.err
  python - "$out/bench_$w.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r, c = d["roofline"], d["roofline_config"]
print(f"{d['config']['workload'][:34]:36s} config {c['us_per_launch']:8.2f} us/step | HBM-sized {r['batch']:>8d} rows {r['us_per_launch']:8.1f} us  "
      f"{r['evals_per_s'] / 1e9:6.2f} G/s  frac {r['frac']:.3f}")
PY
done
