for v in mw4 mw3; do for w in 4 0; do echo -n "$v wpb=$w "; SNPM_FORCE_WPB=$w timeout -k 10 200 python tools/ab/ab_bits.py tools/ab/libq4_$v.so 10000 50000000 pl 2>&1 | tail -1; done; done
echo -n "mw4 wpb=4 1135x40M "; SNPM_FORCE_WPB=4 timeout -k 10 200 python tools/ab/ab_bits.py tools/ab/libq4_mw4.so 1135 40000000 pl 2>&1 | tail -1
echo -n "p16 1135x40M "; SNPM_P16_Q4=0 timeout -k 10 200 python tools/ab/ab_bits.py tools/ab/libq4_mw4.so 1135 40000000 pl 2>&1 | tail -1
