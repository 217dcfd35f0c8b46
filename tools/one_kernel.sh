#!/bin/bash
# One instantiation of a kernel, compiled alone against a header directory (seconds instead of the library's minutes):
#   tools/one_kernel.sh <out-prefix> '<explicit instantiation>' [hdr-dir] [extra hipcc flags...]
# e.g. tools/one_kernel.sh /tmp/k6 'osc_kernel<StaticArm<Tab_ur5>, double, 6, false, 0, 1, true>' abr_control_amd/csrc -DABRK_MARKS
# -> <out-prefix>.o (tools/kernel_resources.py), <out-prefix>.s (device ISA: tools/phase_counts.py, asm_histogram.py)
set -e
out=$1; inst=$2; hdr=${3:-abr_control_amd/csrc}; shift; shift; shift || true
here=$(cd "$(dirname "$0")/.." && pwd)
case $hdr in /*) ;; *) hdr=$here/$hdr;; esac
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-signed-zeros -ffinite-math-only -mllvm -amdgpu-sched-strategy=max-ilp -fno-slp-vectorize"
cat > $out.hip <<SRC
#include "abrk_kernels.h"
namespace abrk {
template __global__ void $inst(${ARGS:-StaticArm<Tab_ur5>, OscP<double>, long, const double*, const double*, const double*, const double*, double*, const double*, double*, double*, int, int*, double*});
}
SRC
/opt/rocm/bin/hipcc $flags -I$hdr -I$here/include "$@" -c $out.hip -o $out.o
/opt/rocm/bin/hipcc $flags -I$hdr -I$here/include "$@" -S --cuda-device-only $out.hip -o $out.s
python3 $here/tools/kernel_resources.py $out.o | grep -v "^$" | head -5
