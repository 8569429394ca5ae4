#!/bin/bash
# k_fast_packed_q4 on narrow packed panels: helper waves (build tables only) x tile rows, same box
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04p; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-30s %6s x %9s  %-18s %.3f ms  frac %.4f  top_ok %s' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['frac'], d['checks']['top_hit_is_planted']))"; }
{
for shape in "1135 40000000" "1024 40000000" "512 100000000" "2029 30000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end --no-real-panel"
  for cfg in "0 0" "1 0" "2 0" "3 0" "2 32" "2 64" "0 64" "6 64"; do
    set -- $shape $cfg
    SNPM_Q4_HELPER_WAVES=$3 SNPM_Q4_TILE_ROWS=$4 timeout -k 10 200 python bench.py --packed $common 2>/dev/null | line "helpers=$3 tile_rows=$4" $1 $2
  done
done
} | tee $out/q4_helper_waves.txt
