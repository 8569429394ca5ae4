#!/bin/bash
# Builds libsnpmatch_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -euo pipefail
cd "$(dirname "$0")"
mkdir -p build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off \
    -pthread -Iinclude -Isnpmatch_amd/csrc ${SNPM_SAVE_TEMPS:+-save-temps=obj} "$@" \
    -o build/libsnpmatch_hip.so snpmatch_amd/csrc/snpm_api.hip snpmatch_amd/csrc/snpm_vcf.cpp snpmatch_amd/csrc/snpm_host.cpp snpmatch_amd/csrc/snpm_h5.cpp -lz
cp build/libsnpmatch_hip.so snpmatch_amd/libsnpmatch_hip.so
