#!/bin/bash
# the split layout of packed panels against round 3's whole rows (SNPM_PACKED_SPLIT=0): dense scans over panel widths, the
# request-size-split PMC of the 1135-accession scans, the packed real-panel legs
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04m; mkdir -p $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-18s %6s x %9s  %-18s %.3f ms  %.0f GB/s  frac %.4f  pitch %s' % ('$1','$2','$3', r['kernel'], r['avg_ms'], r['achieved'], r['frac'], d['config']['workload'].split('= ')[-1].split(',')[0]))"; }
{
for shape in "128 200000000" "256 100000000" "512 100000000" "1040 40000000" "1135 40000000" "1135 11000000" "1300 40000000" "2100 30000000" "3000 20000000" "4400 12000000" "5000 12000000"; do
  set -- $shape
  common="--n-acc $1 --n-snp $2 --mode fast --steps 6 --warmup 2 --no-cpu-baseline --no-alternatives --no-end-to-end --no-real-panel"
  for split in 0 1; do
    SNPM_PACKED_SPLIT=$split timeout -k 10 200 python bench.py --packed --hard-calls $common 2>/dev/null | line "hard split=$split" $1 $2
    SNPM_PACKED_SPLIT=$split timeout -k 10 200 python bench.py --packed $common 2>/dev/null | line "PL   split=$split" $1 $2
  done
done
} | tee $out/split_layout_sweep.txt
A="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum"
for kind in bits q4; do
  hard=0; [ $kind = bits ] && hard=1
  ( export PMC_N_ACC=1135 PMC_N_SNP=40000000 PMC_PACKED=1 PMC_HARD=$hard; rocprofv3 --pmc $A --output-format csv -d $out/pmc_$kind -- python3 tools/pmc_workload_dense.py > $out/pmc_$kind.log 2>&1 )
  python tools/pmc_split.py $out/pmc_$kind > $out/pmc_split_${kind}_1135x40M_split_layout.json
  rm -rf $out/pmc_$kind
  python - <<PY
import json
for r in json.load(open("$out/pmc_split_${kind}_1135x40M_split_layout.json"))[:3]:
    print("%-6s %-60s n=%-3d read %.4g B per launch (128-B requests %.3g)" % ("$kind", r["kernel"][:60], r["launches"], r.get("read_bytes_per_launch") or 0, r.get("TCC_EA0_RDREQ_128B_sum") or 0))
PY
done | tee $out/pmc_split_layout_summary.txt
timeout -k 10 300 python tools/bench_real_panel.py --formats packed --reps 20 > $out/real_panel_packed_split.json 2>/dev/null
SNPM_PACKED_SPLIT=0 timeout -k 10 300 python tools/bench_real_panel.py --formats packed --reps 20 > $out/real_panel_packed_whole_rows.json 2>/dev/null
python - <<'PY'
import json
for f in ("real_panel_packed_split", "real_panel_packed_whole_rows"):
    d = json.load(open("gpurun_out/r04m/%s.json" % f))
    v = d["formats"]["packed"]
    print(f, "pitch", v["row_pitch"], "panel %.2f GB" % v["panel_gb"])
    for l in v["legs"]:
        print("   %-44s wall %.3f ms kernel %.4f ms frac %.3f" % (l["leg"], l["wall_ms_per_call"], l["kernel_ms_per_call"], l["frac_of_hbm_peak"]))
PY
