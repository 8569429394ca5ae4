#!/bin/bash
# A/B of the LDS-bank swizzle of k_fast_packed_q4 (SNPM_Q4_SWZ = 0 / 1 / 2 builds under tools/ab/): whole packed job and two other
# widths, then the LDS counters of every variant.  Run on the GPU box from the repo root.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03/q4_swz; mkdir -p $out
for v in 0 1 2 0 1 2; do
  for shape in "10000 50000000" "1135 40000000" "5000 50000000"; do
    set -- $shape
    SNPMATCH_HIP_LIB=$PWD/tools/ab/libq4_swz$v.so timeout -k 10 200 python bench.py --packed --n-acc $1 --n-snp $2 --steps 8 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('swz=$v  %s x %s  kernel %.3f ms  step %.3f ms  frac(packed bytes) %.3f  checks %s' % ('$1','$2', r['avg_ms'], d['ms_per_step'], r['frac'], d['checks']['top_hit_is_planted']))"
  done
done | tee $out/ab.txt
export PMC_N_SNP=2000000 PMC_PACKED=1
for v in 0 1 2; do
  rm -rf $out/sq$v; mkdir -p $out/sq$v
  i=0
  for s in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"; do
    SNPMATCH_HIP_LIB=$PWD/tools/ab/libq4_swz$v.so timeout -k 10 200 rocprofv3 --pmc $s --output-format csv -d $out/sq$v/p$i -- python3 tools/pmc_sq_run.py > $out/sq$v/p$i.log 2>&1
    echo "swz $v pass $i rc=$?"
    i=$((i+1))
  done
  python3 tools/pmc_sq_summary.py $out/sq$v k_fast_packed_q4 > $out/sq_swz$v.txt 2>&1
  tail -16 $out/sq_swz$v.txt
done
find $out -name "*.db" -delete 2>/dev/null
echo done
