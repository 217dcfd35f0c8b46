#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
bash tools/gpu_sanitize.sh $O/sanitizers
