#!/bin/bash
# Round 6, seventh GPU call: GPU suite on the build with every input of the general-chain first pass requested up front;
# A/B of the record branch reading all six task rows before its first store (r1) against the shipped form (final6).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6i; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
bash tools/gpu_r6_ab.sh r6i/rec 4096,16384,65536,8388608 3 final6 r1
S="--steps 400 --warmup 50 --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
timeout 300 python bench.py --workload osc5_j2 $S > $O/bench_osc5_j2.json 2> $O/bench_osc5_j2.err
python -c "
import json; d=json.loads(open('$O/bench_osc5_j2.json').read().strip().splitlines()[-1]); print('osc5_j2 step', d['roofline_config']['us_per_launch'], '8M', d['roofline']['us_per_launch'], d['roofline']['frac'])"
