L=snpmatch_amd/libsnpmatch_hip.so
for w in 0 4 8; do echo -n "q4=1 wpb=$w  "; SNPM_FORCE_WPB=$w timeout -k 10 200 python tools/ab/ab_bits.py $L 10000 50000000 pl 2>&1 | tail -1; done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py -x -q -k "packed" 2>&1 | tail -3
