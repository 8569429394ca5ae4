#!/bin/bash
# SNPM_OCC_CAP=2 against the default on shapes whose column blocks are full 8-wave blocks (n_acc a multiple of 2048 or close below)
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03b; mkdir -p $out
for shape in "2048 50000000" "4096 40000000" "6144 30000000" "8192 24000000" "16384 12000000" "10000 5000000" "10000 1000000" "2048 8000000" "20480 9000000" "10000 20019000"; do
  set -- $shape
  for cap in 0 2; do
    SNPM_OCC_CAP=$cap timeout -k 10 200 python bench.py --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cap=$cap  %6s x %9s  kernel %.3f ms  frac %.4f' % ('$1','$2', r['avg_ms'], r['frac']))"
  done
done | tee $out/ab_occ_cap2.txt
