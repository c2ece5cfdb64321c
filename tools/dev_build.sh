#!/bin/bash
# Developer loop: rebuild the gfx950 library (+ resource usage of the six-wave pair kernel) and the CPU-emulation twin.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/shoeprint-image-retrieval_amd/csrc
mkdir -p $C/build
( cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -Wall -Wno-unused-function \
    -Rpass-analysis=kernel-resource-usage -c ncc_pair6.hip -o build/ncc_pair6.o 2>&1 | grep -E "error|VGPRs|Scratch|Occupancy|SGPRs:" || true )
make -C $C -j8 2>&1 | grep -E "error|Error" && exit 1 || true
[ "$1" = "noemu" ] || python $R/tests/emu/build_emu.py > /dev/null
echo "build ok"
