#!/bin/bash
# usage: tools/build_variant.sh NAME -DMACRO=1 ...   -> tools/ab/libsnpmatch_hip_NAME.so (select with SNPMATCH_HIP_LIB)
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/ab
${HIPCC:-/opt/rocm/bin/hipcc} -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off \
    -pthread -Iinclude -Isnpmatch_amd/csrc -DSNPM_BUILD_ID=\"variant-$name\" "$@" \
    -o tools/ab/libsnpmatch_hip_$name.so snpmatch_amd/csrc/snpm_api.hip snpmatch_amd/csrc/snpm_vcf.cpp snpmatch_amd/csrc/snpm_host.cpp snpmatch_amd/csrc/snpm_h5.cpp -lz
