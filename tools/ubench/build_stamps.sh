#!/bin/bash
# -DSPR_STAMPS build of the library for tools/ubench/stamps_pair6.py (diagnostic only)
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/shoeprint-image-retrieval_amd/csrc
cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -Wall -Wno-unused-function -DSPR_STAMPS -c ncc_pair6.hip -o /tmp/pair6_stamps.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v ncc_pair6.o) /tmp/pair6_stamps.o -o $R/tools/ubench/libstamps.so && echo stamps ok
