#!/bin/bash
# row tiles of the shared-row contraction (SNPM_SHARED_TILES) against its kernel time -> gpurun_out/$1/tiles_sweep.txt
set -uo pipefail
out=gpurun_out/${1:-r05m}; mkdir -p $out
f=$out/tiles_sweep.txt; : > $f
for B in 64 128; do
  for T in 0 24 32 56 64 72; do
    echo "samples=$B tiles=$T (0 = the planner)" >> $f
    SNPM_SHARED_TILES=$T timeout -k 10 200 python tools/time_shared.py $B 200000 10 2>/dev/null | tail -1 | cut -c1-330 >> $f || exit 1
  done
done
cat $f
