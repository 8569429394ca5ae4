#!/bin/bash
# resident blocks per CU of k_fast (SNPM_OCC_CAP) on the per-GPU shapes of the N = 1 / 2 / 4 / 8 runs and the config-5 slab
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03b; mkdir -p $out
for rep in 1 2; do
for shape in "10000 20000000" "5000 40000000" "2500 50000000" "1252 50000000" "12500 16000000" "1135 50000000"; do
  set -- $shape
  for cap in 0 2 3 4; do
    SNPM_OCC_CAP=$cap timeout -k 10 200 python bench.py --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cap=$cap  %6s x %9s  kernel %.3f ms  frac %.4f' % ('$1','$2', r['avg_ms'], r['frac']))"
  done
done
done | tee $out/ab_occ_cap.txt
