#!/bin/bash
# Timing experiment: k_fast_packed_q4 as shipped vs the same kernel without its 4 x 16 transpose of 2-bit fields
# (tools/build_variant.sh q4proto -DSNPM_Q4_PROTO_QUAD=1: what the lookups would cost on a panel stored four rows per
# byte; its results are wrong by construction, only the time is read).
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03h; mkdir -p $out
for shape in "10000 50000000" "8192 50000000" "5000 50000000" "1135 40000000"; do
  set -- $shape
  for lib in "" tools/ab/libsnpmatch_hip_q4proto.so; do
    SNPMATCH_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 200 python bench.py --packed --n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>$out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s %6s x %9s  %s %.3f ms  frac %.4f' % ('${lib:+proto}','$1','$2', r['kernel'], r['avg_ms'], r['frac']))" || tail -5 $out/err.txt
  done
done | tee $out/ab_q4_proto_quad.txt
