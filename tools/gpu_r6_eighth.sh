#!/bin/bash
# Round 6, eighth GPU call: A/B of branch hints on the cold sin/cos fallbacks (block placement moves the library routine
# out of the hot path: denser instruction stream) on the six-row first pass, the x,y,z law and the x,y,z + Coriolis law.
cd $GRAFT_REPO_ROOT
bash tools/gpu_r6_ab.sh r6j/osc6 4096,131072,8388608 2 final6 e1
bash tools/gpu_r6_ab.sh r6j/cfg2 4096,131072,8388608 2 base3 e1c2
bash tools/gpu_r6_ab.sh r6j/cfg4 4096,131072,8388608 2 base3c e1c4
