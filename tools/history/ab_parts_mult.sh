#!/bin/bash
# narrow packed panels with a phased last wave: more parts than resident blocks, so that the slots the short tail blocks
# leave are refilled (SNPM_PARTS_MULT)
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03h; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s %6s x %9s  %-18s %.3f ms  %.0f GB/s  frac %.4f  step %.3f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['achieved'], r['frac'], d['ms_per_step']))"; }
for shape in "1135 40000000" "1135 11000000" "1040 40000000" "2400 20000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end --packed"
  for m in 1 2 4 8; do
    SNPM_PARTS_MULT=$m timeout -k 10 200 python bench.py --hard-calls $common 2>/dev/null | line hard-mult$m $1 $2
    SNPM_PARTS_MULT=$m timeout -k 10 200 python bench.py $common 2>/dev/null | line PL-mult$m $1 $2
  done
done | tee $out/ab_parts_mult.txt
