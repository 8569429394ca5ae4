#!/bin/bash
# k_fast_packed_q4: more parts than resident blocks (SNPM_PARTS_MULT), by panel width
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03j; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-10s %6s x %9s  %-18s %.3f ms  frac %.4f  step %.3f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['frac'], d['ms_per_step']))"; }
for shape in "512 100000000" "1135 40000000" "2029 30000000" "2400 20000000" "3000 20000000" "4096 20000000" "5000 20000000" "8192 20000000" "10000 20000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end --packed"
  for m in 1 2 4 8; do
    SNPM_PARTS_MULT=$m timeout -k 10 200 python bench.py $common 2>/dev/null | line mult=$m $1 $2
  done
done | tee $out/ab_q4_parts_mult.txt
