#!/bin/bash
# same-box A/B of k_fast_packed_q4's grid order on wide packed panels: (column blocks, parts) as shipped vs (parts, column blocks)
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04r; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s %6s x %9s  %-18s %.3f ms  frac %.4f  ok %s' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['frac'], d['checks']['top_hit_is_planted']))"; }
{
for shape in "2029 30000000" "3000 20000000" "4096 20000000" "5000 20000000" "8192 20000000" "10000 20000000" "16384 10000000" "20000 10000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end --no-real-panel"
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --packed $common 2>/dev/null | line "colblock-fast" $1 $2
    SNPMATCH_HIP_LIB=$PWD/tools/ab/libsnpmatch_hip_q4pf.so timeout -k 10 200 python bench.py --packed $common 2>/dev/null | line "part-fast" $1 $2
  done
done
} | tee $out/q4_grid_order_ab.txt
