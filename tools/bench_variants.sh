#!/bin/bash
# usage: bench_variants.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each arg = one bench run with that environment)
for v in "$@"; do
  out=$(env $v timeout -k 10 200 python "$(dirname $0)/../bench.py" --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null)
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s value %.3e  k_fast %.0f GB/s (%.3f)  avg %.3f ms  step %.3f ms  ok=%s' % ('$v', d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_ms'], d['ms_per_step'], d['checks']['top_hit_is_planted']))"
done
