#!/bin/bash
# k_fast_bits / k_fast_packed_q4 around one wave of accessions: is the nearly empty second wave of a 1135-wide panel the cost?
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03h; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s %6s x %9s  %-18s %.3f ms  %.0f GB/s  frac %.4f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['achieved'], r['frac']))"; }
for shape in "1024 40000000" "1040 40000000" "1135 40000000" "1536 40000000" "2048 40000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end --packed"
  timeout -k 10 200 python bench.py --hard-calls $common 2>/dev/null | line hard $1 $2
  SNPM_PITCH_ALIGN=64 timeout -k 10 200 python bench.py --hard-calls $common 2>/dev/null | line hard-pitch64 $1 $2
  timeout -k 10 200 python bench.py $common 2>/dev/null | line PL $1 $2
  SNPM_PITCH_ALIGN=64 timeout -k 10 200 python bench.py $common 2>/dev/null | line PL-pitch64 $1 $2
done | tee $out/ab_bits_narrow.txt
