#!/bin/bash
# request-size-split PMC passes (tools/pmc_split.py): dense narrow panels and the gathered real-panel legs -> gpurun_out/r04e/
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04e; mkdir -p $out
A="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum"
B="FETCH_SIZE"
pass() {   # name, env..., -- program
  name=$1; shift
  for set in A B; do
    ctrs=$A; [ $set = B ] && ctrs=$B
    env "$@" true
    ( export "$@"; rocprofv3 --pmc $ctrs --output-format csv -d $out/${name}_$set -- python3 $PROG > $out/${name}_$set.log 2>&1 )
  done
  python tools/pmc_split.py $out/${name}_A $out/${name}_B > $out/pmc_split_$name.json
  rm -rf $out/${name}_A $out/${name}_B
  python - <<PY
import json
for r in json.load(open("$out/pmc_split_$name.json"))[:6]:
    print("%-14s %-60s grid %-10s n=%-3d read %.4g B (32B %.3g 64B %.3g 128B %.3g; classes/all %.3f)  FETCH_SIZE raw %.4g" % ("$name", r["kernel"][:60], r["grid"], r["launches"], r.get("read_bytes_per_launch") or 0, r.get("TCC_EA0_RDREQ_32B_sum") or 0, r.get("TCC_EA0_RDREQ_64B_sum") or 0, r.get("TCC_EA0_RDREQ_128B_sum") or 0, r.get("classes_over_all_requests") or 0, r.get("FETCH_SIZE_bytes_raw") or 0))
PY
}
PROG=tools/pmc_workload_dense.py
pass int8_10000x6M PMC_N_ACC=10000 PMC_N_SNP=6250000 PMC_PACKED=0
pass int8_1135x40M PMC_N_ACC=1135 PMC_N_SNP=40000000 PMC_PACKED=0
pass bits_1135x40M PMC_N_ACC=1135 PMC_N_SNP=40000000 PMC_PACKED=1 PMC_HARD=1
pass q4_1135x40M PMC_N_ACC=1135 PMC_N_SNP=40000000 PMC_PACKED=1 PMC_HARD=0
PROG="tools/bench_real_panel.py --formats int8 --reps 2"
pass real_int8 X=1
PROG="tools/bench_real_panel.py --formats packed --reps 2"
pass real_packed X=1
