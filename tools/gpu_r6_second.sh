#!/bin/bash
# Round 6, second GPU call: the GPU suite again (first call stopped at the rollout leg of the bench-workload test), the
# single-process bench line.  -> gpurun_out/r6b/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6b; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 600 python bench.py --gpus 8 --single-process --allow-shared-device --steps 20 --warmup 5 > $O/bench_sp8.json 2> $O/bench_sp8.err
tail -c 1500 $O/bench_sp8.json
