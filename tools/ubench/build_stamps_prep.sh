#!/bin/bash
# -DSPR_PREP_STAMPS build of the library for tools/ubench/stamps_prep.py (diagnostic only)
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/shoeprint-image-retrieval_amd/csrc
cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -Wall -Wno-unused-function -DSPR_PREP_STAMPS -c ncc_fft.hip -o /tmp/fft_stamps.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v ncc_fft.o) /tmp/fft_stamps.o -o $R/tools/ubench/libstamps_prep.so && echo stamps ok
