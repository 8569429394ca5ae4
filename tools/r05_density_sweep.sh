#!/bin/bash
# break-even of the shared-row scan against the per-sample pass by density (share of the marker set a sample has calls at)
# -> gpurun_out/$1/density_sweep.txt; the automatic policy's thresholds (snpm_api.hip: shared_min_density_of) come from here
set -uo pipefail
out=gpurun_out/${1:-r05m}; mkdir -p $out
f=$out/density_sweep.txt; : > $f
for packed in 0 1; do for drop in 0.6 0.7 0.8 0.85 0.9 0.95; do
  echo "-- packed=$packed, each of 64 samples lacks $drop of the 200k-marker set" >> $f
  timeout -k 10 200 python tools/time_shared.py 64 200000 6 0 $packed $drop 2>/dev/null | grep "pass\|scan" | cut -c1-75 >> $f || exit 1
done; done
cat $f
