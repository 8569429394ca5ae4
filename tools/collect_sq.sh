#!/bin/bash
# SQ counters of the reference-order kernel (int8 and packed panels) and of the packed fast pass: separate --pmc passes of
# four counters each; summaries under gpurun_out/r02b/sq_*.txt.   Run on the GPU box from the repo root.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02b; mkdir -p $out
export PMC_N_SNP=2000000
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES")
run() {   # name, kernel substring, script
  rm -rf $out/sq_$1; mkdir -p $out/sq_$1
  i=0
  for s in "${sets[@]}"; do
    timeout -k 10 200 rocprofv3 --pmc $s --output-format csv -d $out/sq_$1/p$i -- python3 $3 > $out/sq_$1/p$i.log 2>&1
    echo "$1 pass $i rc=$?"
    i=$((i+1))
  done
  python3 tools/pmc_sq_summary.py $out/sq_$1 "$2" > $out/sq_$1.txt 2>&1
  tail -22 $out/sq_$1.txt
}
export PMC_PACKED=0; run strict_int8 k_strict4 tools/pmc_sq_strict.py
export PMC_PACKED=1; run strict_packed k_strict4 tools/pmc_sq_strict.py
export PMC_PACKED=1; run fast_packed_q4 k_fast_packed_q4 tools/pmc_sq_run.py
find $out -name "*.db" -delete 2>/dev/null
echo done
