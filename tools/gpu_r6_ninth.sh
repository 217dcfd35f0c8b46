#!/bin/bash
# Round 6, ninth GPU call: A/B of -mllvm -disable-machine-licm (no hoisting of literals out of the persistent loops: the
# hoisted ones overflow the scalar registers and come back through v_readlane / v_accvgpr_read, each a vector-ALU slot).
cd $GRAFT_REPO_ROOT
bash tools/gpu_r6_ab.sh r6n/osc6 48,4096,65536,1048576,8388608 2 final6 m6
bash tools/gpu_r6_ab.sh r6n/j2 4096,8388608 2 j1 m6j
bash tools/gpu_r6_ab.sh r6n/cfg2 4096,8388608 2 base3 m3
bash tools/gpu_r6_ab.sh r6n/cfg4 4096,8388608 2 base3c m3c
