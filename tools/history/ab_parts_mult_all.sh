#!/bin/bash
# more parts than resident blocks (SNPM_PARTS_MULT): the three fast kernels on their main shapes
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03j; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s %6s x %9s  %-18s %.3f ms  frac %.4f  step %.3f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['frac'], d['ms_per_step']))"; }
for shape in "10000 20000000" "8192 20000000" "1135 40000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end"
  for m in 8 16 32; do
    SNPM_PARTS_MULT=$m timeout -k 10 200 python bench.py --packed $common 2>/dev/null | line q4-mult=$m $1 $2
  done
  for m in 1 2 4 8; do
    SNPM_PARTS_MULT=$m timeout -k 10 200 python bench.py --packed --hard-calls $common 2>/dev/null | line bits-mult=$m $1 $2
  done
done
for shape in "10000 20000000" "1250 50000000" "2500 50000000" "5000 40000000" "1135 40000000" "12500 16000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end"
  for m in 1 2 4 8; do
    SNPM_PARTS_MULT=$m timeout -k 10 200 python bench.py $common 2>/dev/null | line int8-mult=$m $1 $2
  done
done | tee -a /dev/null
