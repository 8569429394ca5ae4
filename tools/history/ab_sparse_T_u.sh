#!/bin/bash
# rows in flight per lane of k_strict_sparse_T (8 as shipped in round 2, 16, 32): the re-evaluation of one flagged accession over 50M rows
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03h; mkdir -p $out
for lib in "" tools/ab/libsnpmatch_hip_sparseU16.so tools/ab/libsnpmatch_hip_sparseU32.so; do
  for args in "--n-acc 1250 --n-snp 50000000" "--packed"; do
    SNPM_SPARSE_T_DENSE=0 SNPMATCH_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 400 python bench.py $args --steps 10 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s %-34s step %.3f ms  kernel %.3f ms  (step - kernel %.3f ms)  reeval %s' % ('${lib:-default (U=8)}', '$args', d['ms_per_step'], r['avg_ms'], d['ms_per_step']-r['avg_ms'], d['checks']['strict_reevaluations']))"
  done
done | tee $out/ab_sparse_T_u.txt
