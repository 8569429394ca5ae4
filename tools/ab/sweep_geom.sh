# geometry sensitivity: waves per block x parts multiplier, full build
L=snpmatch_amd/libsnpmatch_hip.so
echo "## k_fast_bits"
for w in 1 2 3; do for m in 1 2 4; do
  echo -n "wpb=$w mult=$m  "; SNPM_FORCE_WPB=$w SNPM_PARTS_MULT=$m timeout -k 10 200 python tools/ab/ab_bits.py $L 2>&1 | tail -1
done; done
echo "## k_fast_packed_q4"
for w in 0 2 3 4; do for m in 1 2; do
  echo -n "wpb=$w mult=$m  "; SNPM_FORCE_WPB=$w SNPM_PARTS_MULT=$m timeout -k 10 200 python tools/ab/ab_bits.py $L 10000 50000000 pl 2>&1 | tail -1
done; done
echo "## k_fast int8 10000 x 20M"
for w in 0 2 3 4 8; do for m in 1 2; do
  echo -n "wpb=$w mult=$m  "; SNPM_FORCE_WPB=$w SNPM_PARTS_MULT=$m timeout -k 10 200 python tools/ab/ab_fast.py $L 2>&1 | tail -1
done; done
