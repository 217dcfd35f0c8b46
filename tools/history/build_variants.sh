#!/bin/bash
# Kernel-variant experiment: rebuild only the UR5 translation unit with different launch
# bounds / block sizes and link each into its own libabrk_<tag>.so (selected at run time with
# ABRK_LIB_PATH).  Usage: tools/build_variants.sh "w1b64:-DABRK_MIN_WAVES=1 -DABRK_BLOCK=64" ...
set -e
cd "$(dirname "$0")/../abr_control_amd/csrc"
mkdir -p variants
TU=${TU:-ur5}   # which arm translation unit to rebuild (TU=jaco2 tools/build_variants.sh ...)
OTHERS=$(ls build/*.o | grep -v abrk_arm_$TU.o)
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-signed-zeros -ffinite-math-only -mllvm -amdgpu-sched-strategy=max-ilp -fno-slp-vectorize $flags -c abrk_arm_$TU.hip -o variants/${TU}_$tag.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libabrk_$tag.so variants/${TU}_$tag.o $OTHERS \
    && echo "built $tag" ) &
done
wait
