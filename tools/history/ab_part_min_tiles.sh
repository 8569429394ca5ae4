#!/bin/bash
# k_fast_packed_q4's parts rule: tiles (256 rows) a part keeps (SNPM_PART_MIN_TILES), judged by the time of the whole step
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03j; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s %6s x %9s  kernel %.3f ms  frac %.4f  step %.3f ms' % ('$1','$2','$3', r['avg_ms'], r['frac'], d['ms_per_step']))"; }
for shape in "1135 40000000" "1135 11000000" "2400 20000000" "3000 20000000" "10000 20000000" "10000 6250000" "8192 8000000" "512 100000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 8 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end --packed"
  for t in 32 16 8 4; do
    SNPM_PART_MIN_TILES=$t timeout -k 10 300 python bench.py $common 2>/dev/null | line min=$t $1 $2
  done
done | tee $out/ab_part_min_tiles.txt
