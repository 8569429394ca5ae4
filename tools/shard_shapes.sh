#!/bin/bash
# what one GPU of an N-GPU run of the whole job does: the accession shard of 10 000 / N accessions x 50M SNPs, exact mode
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03h; mkdir -p $out
for shape in "1250 8" "2500 4" "5000 2"; do
  set -- $shape
  timeout -k 10 400 python bench.py --n-acc $1 --n-snp 50000000 --steps 10 --warmup 2 --no-cpu-baseline --no-alternatives > $out/bench_shard_${1}x50M_n$2.json 2>$out/bench_shard_$1.err
  python - <<PY
import json
d=json.loads(open("$out/bench_shard_${1}x50M_n$2.json").read().strip().splitlines()[-1]); r=d['roofline']
print("%5d x 50M (one of $2 GPUs): step %.3f ms  k_fast %.3f ms  frac %.4f  e2e frac %s  reeval %s  slabs %s  -> N=$2 job value ~ %.3e" % ($1, d['ms_per_step'], r['avg_ms'], r['frac'], r.get('end_to_end_frac'), d['checks']['strict_reevaluations'], d['config']['slabs'], 1e4*5e7/(d['ms_per_step']*1e-3)))
PY
done | tee $out/shard_shapes.txt
