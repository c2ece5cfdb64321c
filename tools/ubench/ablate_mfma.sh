#!/bin/bash
# Timing ablations of the matrix-core pair kernel (wrong results, timing only): builds tools/ubench/libabl_mfma_<n>.so
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/shoeprint-image-retrieval_amd/csrc
for n in "$@"; do
  ( cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -Wall -Wno-unused-function -fno-slp-vectorize -DSPR_MFMA_ABL=$n -c ncc_mfma.hip -o /tmp/mfma_abl_$n.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v ncc_mfma.o) /tmp/mfma_abl_$n.o -o $R/tools/ubench/libabl_mfma_$n.so ) && echo "ablation $n built"
done
