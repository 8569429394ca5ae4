#!/bin/bash
# usage: bench_shape.sh N_ACC N_SNP "ENV=.." ...   one fast-mode bench run of that (single-GPU) shape per environment string
nacc=$1; snps=$2; shift 2
for v in "$@"; do
  out=$(env $v timeout -k 10 300 python "$(dirname $0)/../bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --mode fast --n-acc $nacc --n-snp $snps 2>/dev/null)
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%6d x %9d %-28s k_fast %.0f GB/s (%.3f)  avg %.3f ms  step %.3f ms' % ($nacc, $snps, '$v', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_ms'], d['ms_per_step']))"
done
