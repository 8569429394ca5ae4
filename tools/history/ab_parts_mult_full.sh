#!/bin/bash
# the whole 10 000 x 50M packed job: parts per resident block
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03j; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-14s %-18s %.3f ms  frac %.4f  step %.3f ms' % ('$1', r['kernel'], r['avg_ms'], r['frac'], d['ms_per_step']))"; }
for m in 2 4 8 12 16 24; do
  SNPM_PARTS_MULT=$m timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --packed --mode fast 2>/dev/null | line q4-mult=$m
done
for m in 2 3 4; do
  SNPM_PARTS_MULT=$m timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --packed --hard-calls --mode fast 2>/dev/null | line bits-mult=$m
done
