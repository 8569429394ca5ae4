#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04x; mkdir -p $out
{
for env in "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "AMD_DIRECT_DISPATCH=0" "GPU_MAX_COMMAND_BUFFERS=64" "DEBUG_CLR_MAX_BATCH_SIZE=1" "GPU_MAX_HW_QUEUES=1" "HSA_ENABLE_INTERRUPT=0"; do
  echo "== ${env:-default}"
  env $env timeout -k 10 200 python tools/debug/r04_stall_probe.py 2>&1 | grep -v amdgpu.ids
done
} | tee $out/stall_probe.txt
