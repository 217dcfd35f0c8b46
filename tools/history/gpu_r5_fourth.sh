#!/bin/bash
# Round 5, fourth GPU call: the early-exit eigen-solver of the six-row law's truncating pseudo-inverse - six-row GPU tests,
# then the steps at every batch size and the HBM-sized leg.  -> gpurun_out/r5d/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -q -k "six_row or handover or controllers_match or fuzz_osc or truncated or Mx or runtime_table or fp32_kernels") > $O/pytest_six.log 2>&1
grep -E "passed|failed|error" $O/pytest_six.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_six.log | head
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
: > $O/ab.txt
for b in 4096 8192 16384 32768 65536; do
  for rep in 1 2; do
  timeout 300 python bench.py --workload osc6 --batch $b $S 2> $O/err_$b.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('osc6 B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt
  done
done
for b in 16384 65536; do timeout 300 python bench.py --workload osc5_j2 --batch $b $S 2>> $O/err_j2.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('osc5_j2 B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt; done
timeout 300 python bench.py --workload osc5_j2 --batch 4096 $S 2> $O/err_j2.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('osc5_j2 B=4096', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt
timeout 300 python bench.py --workload osc6 --steps 200 --warmup 20 --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras > $O/bench_osc6.json 2> $O/bench_osc6.err
python -c "
import json; d=json.loads(open('$O/bench_osc6.json').read().strip().splitlines()[-1]); r=d['roofline']; print('osc6 8M', r['us_per_launch'], r['frac'])"
