#!/bin/bash
# Round 6, third GPU call: the tests the first two calls left open, then same-box A/Bs (six-row first pass: table sin/cos for
# the target angles, EE frame at compile time; x,y,z law at three waves per SIMD).  -> gpurun_out/r6c/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
(timeout 900 python -m pytest tests -m gpu -q -k "resident or merged or bench_single or singular or bench_workloads" ) > gpurun_out/r6c/pytest_sel.log 2>&1; tail -3 gpurun_out/r6c/pytest_sel.log
bash tools/gpu_r6_ab.sh r6c/osc6 4096,16384,65536,8388608 2 base6 v1 v2 v3
bash tools/gpu_r6_ab.sh r6c/cfg2 4096,131072,8388608 2 base3 v4
bash tools/gpu_r6_ab.sh r6c/cfg4 131072,8388608 1 base3c
