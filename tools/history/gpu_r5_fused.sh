#!/bin/bash
# tools/gpu_r5_fused.sh build | run : the fused single-launch prototype of the six-row small-batch step
set -e
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fno-signed-zeros -ffinite-math-only -mllvm -amdgpu-sched-strategy=max-ilp -fno-slp-vectorize -Iabr_control_amd/csrc -Wno-unused-value"
case "$1" in
build) /opt/rocm/bin/hipcc $FLAGS tools/microbench/osc6_fused_proto.hip -o tools/microbench/osc6_fused_proto.bin ;;
run)
  O=gpurun_out/r5_fused; mkdir -p $O
  timeout 120 tools/microbench/osc6_fused_proto.bin $O/fused_1.json; echo rc=$?
  timeout 120 tools/microbench/osc6_fused_proto.bin $O/fused_2.json; echo rc=$?
  cat $O/fused_1.json $O/fused_2.json ;;
esac
