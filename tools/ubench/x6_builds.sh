#!/bin/bash
# A/B builds of the six-wave pair kernel's layout experiments (-DSPR_X6=n) -> tools/ubench/libx6_<n>.so
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/shoeprint-image-retrieval_amd/csrc
cd $C
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -Wall -Wno-unused-function -DSPR_X6=$n -c ncc_pair6.hip -o /tmp/pair6_x6_$n.o 2>&1 | grep -E "error"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v ncc_pair6.o) /tmp/pair6_x6_$n.o -o $R/tools/ubench/libx6_$n.so
done
echo x6 builds ok
