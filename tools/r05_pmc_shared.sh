#!/bin/bash
# SQ counters of the shared-row batch scan's kernels (tools/time_shared.py), one pass -> gpurun_out/$1/shared_pmc_<args>.txt
# usage: tools/r05_pmc_shared.sh [outdir] [time_shared.py arguments ...]
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r05b}; shift || true
mkdir -p $out
args="${*:-64 200000 3}"
tag=$(echo $args | tr ' ' '_')
# SNPM_PMC picks the pass: the SQ set (default), "l2" = hit / miss of the L2, "fetch" / "write" = bytes past the L2 (FETCH_SIZE in
# KiB and on gfx950 HALF the bytes of wide streaming reads: MI355X_MICROARCH.md)
case "${SNPM_PMC:-sq}" in
  l2) ctrs="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" ;;
  fetch) ctrs="FETCH_SIZE" ;;
  write) ctrs="WRITE_SIZE" ;;
  *) ctrs="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" ;;
esac
tag=${tag}_${SNPM_PMC:-sq}
timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc_$tag -- python3 tools/time_shared.py $args > $out/pmc_shared_$tag.log 2> $out/pmc_shared_$tag.err
echo "rc=$?"
f=$(find $out/pmc_$tag -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 - "$f" > $out/shared_pmc_$tag.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "k_sh_" not in k:
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        v.sort()
        print("   %-28s median %14.0f  (n=%d)" % (c, v[len(v) // 2], len(v)))
PY
rm -rf $out/pmc_$tag
cat $out/shared_pmc_$tag.txt
