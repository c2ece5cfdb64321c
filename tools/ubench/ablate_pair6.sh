#!/bin/bash
# Timing ablations of the six-wave pair kernel (-DSPR_ABL=n builds: wrong results, diagnostic only) -> tools/ubench/libabl<n>.so
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/shoeprint-image-retrieval_amd/csrc
cd $C
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -Wall -Wno-unused-function -DSPR_ABL=$n -c ncc_pair6.hip -o /tmp/pair6_abl$n.o 2>&1 | grep -E "error" 
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v ncc_pair6.o) /tmp/pair6_abl$n.o -o $R/tools/ubench/libabl$n.so
done
echo ablation builds ok
