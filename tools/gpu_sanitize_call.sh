#!/bin/bash
# Build container side of the sanitizer job: build the three variants of the host layer, let them travel for ONE gpurun
# call (they are gpurun-ignored otherwise: 42 MB each), run tools/gpu_sanitize.sh on the GPU box, restore the ignore list.
set -e
cd "$(dirname "$0")/.."
make -C abr_control_amd/csrc asan tsan ubsan > /tmp/make_san.log 2>&1
cp .gpurunignore /tmp/gpurunignore.keep
grep -v "libabrk_.*san\.so" /tmp/gpurunignore.keep > .gpurunignore
trap 'cp /tmp/gpurunignore.keep .gpurunignore' EXIT
/usr/local/graft/bin/gpurun --timeout ${1:-1500} -- 'bash tools/gpu_sanitize.sh gpurun_out/sanitizers'
