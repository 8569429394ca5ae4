#!/bin/bash
# row pitches that are powers of two (8192 / 16384 accessions int8) against the same rows padded by 256 B
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03k; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-26s %6s x %9s  %-18s %.3f ms  frac %.4f' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['frac']))"; }
run() { # n_acc n_snp align fmt
  SNPM_PITCH_ALIGN=$3 timeout -k 10 200 python bench.py $4 --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | line "pitch-align=$3 $4" $1 $2
}
run 8192 20000000 256 ""; run 8192 20000000 8448 ""; run 8192 20000000 8320 ""
run 16384 12000000 256 ""; run 16384 12000000 16640 ""
run 4096 40000000 256 ""; run 4096 40000000 4352 ""
run 2048 50000000 256 ""; run 2048 50000000 2304 ""
run 32768 20000000 256 "--packed"; run 32768 20000000 8448 "--packed"
run 16384 30000000 256 "--packed"; run 16384 30000000 4352 "--packed"
run 32768 20000000 256 "--packed --hard-calls"; run 32768 20000000 8448 "--packed --hard-calls"
