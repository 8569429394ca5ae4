#!/bin/bash
# usage: collect_pmc.sh KEY N_ACC N_SNP   (run on the GPU box from the repo root; writes gpurun_out/pmc_KEY/*)
set -euo pipefail
key=$1; export PMC_N_ACC=$2; export PMC_N_SNP=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$key
mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/pmc_run.py > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/pmc_run.py > $out/write.log 2>&1
cp profiles/pmc_traffic.json $out/pmc_traffic.json 2>/dev/null || true
python tools/pmc_traffic.py $out/fetch $out/write $key $out/pmc_traffic.json | tail -4
