#!/bin/bash
# Builds libsnpmatch_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -euo pipefail
cd "$(dirname "$0")"
mkdir -p build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# build id = hash of every source the library is made of: measurements kept in profiles/ carry it, and bench.py drops a
# recorded figure (PMC traffic) that was collected on another build
BUILD_ID=$(cat snpmatch_amd/csrc/*.hip snpmatch_amd/csrc/*.hpp snpmatch_amd/csrc/*.cpp include/*.h | sha1sum | cut -c1-12)
$HIPCC -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off \
    -pthread -Iinclude -Isnpmatch_amd/csrc -DSNPM_BUILD_ID=\"$BUILD_ID\" ${SNPM_SAVE_TEMPS:+-save-temps=obj} "$@" \
    -o build/libsnpmatch_hip.so snpmatch_amd/csrc/snpm_api.hip snpmatch_amd/csrc/snpm_vcf.cpp snpmatch_amd/csrc/snpm_host.cpp snpmatch_amd/csrc/snpm_h5.cpp -lz
cp build/libsnpmatch_hip.so snpmatch_amd/libsnpmatch_hip.so
