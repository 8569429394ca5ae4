# geometry sensitivity of the packed hard-call pass: pattern-only build vs full build
for lib in tools/ab/libsnpmatch_pattern.so snpmatch_amd/libsnpmatch_hip.so; do
for env in "X=1" "SNPM_FORCE_WPB=2" "SNPM_FORCE_WPB=4" "SNPM_FORCE_WPB=8" "SNPM_PARTS_MULT=2" "SNPM_PARTS_MULT=4" "SNPM_NT=0"; do
  echo "== $lib $env"
  env $env timeout -k 10 200 python tools/ab/ab_bits.py $lib 2>&1 | tail -1
done
done
