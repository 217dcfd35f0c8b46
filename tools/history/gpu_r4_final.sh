#!/bin/bash
# Round 4, final build: the whole GPU suite, smoke(), the six-row step at the sizes of DESIGN §2.1 with the shipped
# slots / rounds rules, the bench line as the driver runs it.   -> gpurun_out/r4final/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4final; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc != 0 ] && { echo "GPU suite failed (rc=$rc)"; exit $rc; }
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
: > $O/ab.txt
for b in 4096 8192 16384 32768 65536; do
  timeout 300 python bench.py --workload osc6 --batch $b $S 2> $O/err_$b.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('osc6 B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt
done
timeout 300 python bench.py --workload osc5_j2 --batch 4096 $S 2> $O/err_j2.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('osc5_j2 B=4096', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-streams-leg > $O/bench_k20.json 2> $O/bench_k20.err
python -c "
import json; d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); print('K20 value', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac']); o=d.get('osc6') or {}; print('osc6 8M frac', o.get('frac'), 'step', (o.get('config_sized_step') or {}).get('us_per_step'))"
