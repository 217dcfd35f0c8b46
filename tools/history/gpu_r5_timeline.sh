#!/bin/bash
# tools/gpu_r5_timeline.sh build   - cross-compile tools/microbench/shard_step_timeline.hip (full + light stamps) here
# tools/gpu_r5_timeline.sh run     - on the GPU box (gpurun): both variants, JSON into gpurun_out/r5_timeline/
set -e
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fno-signed-zeros -ffinite-math-only -mllvm -amdgpu-sched-strategy=max-ilp -fno-slp-vectorize -Iabr_control_amd/csrc -Wno-unused-value"
case "$1" in
build)
  /opt/rocm/bin/hipcc $FLAGS -DABRK_TIMELINE tools/microbench/shard_step_timeline.hip -o tools/microbench/shard_step_timeline.bin &
  /opt/rocm/bin/hipcc $FLAGS -DABRK_TIMELINE -DABRK_TIMELINE_LIGHT tools/microbench/shard_step_timeline.hip -o tools/microbench/shard_step_timeline_light.bin &
  wait
  ;;
build6)
  /opt/rocm/bin/hipcc $FLAGS -DABRK_TIMELINE -DABRK_TIMELINE_LIGHT tools/microbench/osc6_step_timeline.hip -o tools/microbench/osc6_step_timeline.bin
  ;;
run6)
  O=gpurun_out/r5_timeline6; mkdir -p $O
  for rep in 1 2; do timeout 120 tools/microbench/osc6_step_timeline.bin $O/osc6_$rep.json; done
  ls -la $O
  ;;
run)
  O=gpurun_out/r5_timeline
  mkdir -p $O
  for rep in 1 2; do
    timeout 120 tools/microbench/shard_step_timeline.bin $O/full_$rep.json 100
    timeout 120 tools/microbench/shard_step_timeline_light.bin $O/light_$rep.json 100
  done
  ls -la $O
  ;;
esac
