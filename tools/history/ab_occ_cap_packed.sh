#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03c; mkdir -p $out
for hard in "" "--hard-calls"; do
for shape in "10000 50000000" "5000 50000000" "1252 50000000"; do
  set -- $shape
  for cap in 0 1 2 3 4 6; do
    SNPM_OCC_CAP=$cap timeout -k 10 200 python bench.py --packed $hard --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('packed $hard cap=$cap  %6s x %9s  %s %.3f ms  frac %.4f' % ('$1','$2', r['kernel'], r['avg_ms'], r['frac']))"
  done
done
done | tee $out/ab_occ_cap_packed.txt
