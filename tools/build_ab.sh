#!/bin/bash
# tools/build_ab.sh <name> <hdr-dir> <kernel_ab -D flags...>  ->  tools/microbench/ab_<name>.bin (gfx950, the library's flags)
# The arm-independent finish kernels (abrk_law.hip) are compiled from the same header directory once per <hdr-dir>
# (cached as <hdr-dir>/.ab_law.o while the headers are unchanged).
set -e
name=$1; hdr=$2; shift; shift
here=$(cd "$(dirname "$0")/.." && pwd)
case $hdr in /*) ;; *) hdr=$here/$hdr;; esac
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-signed-zeros -ffinite-math-only -mllvm -amdgpu-sched-strategy=max-ilp -fno-slp-vectorize"
sum=$( (cat $hdr/abrk_*.h $hdr/abrk_law.hip; echo "$AB_LAW_FLAGS") | md5sum | cut -c1-16)  # AB_LAW_FLAGS: extra flags of the finish kernels' unit
if [ ! -f $hdr/.ab_law_$sum.o ]; then
  rm -f $hdr/.ab_law_*.o
  /opt/rocm/bin/hipcc $F $AB_LAW_FLAGS -I$hdr -I$here/include -c $hdr/abrk_law.hip -o $hdr/.ab_law_$sum.o
fi
/opt/rocm/bin/hipcc $F -I$hdr -I$here/include "$@" -c $here/tools/microbench/kernel_ab.hip -o /tmp/ab_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/ab_$name.o $hdr/.ab_law_$sum.o -o $here/tools/microbench/ab_$name.bin
ls -la $here/tools/microbench/ab_$name.bin | awk '{print $5, $9}'
