L=snpmatch_amd/libsnpmatch_hip.so
timeout -k 10 200 python tools/ab/ab_bits.py $L 10000 50000000 pl 2>&1 | tail -1
timeout -k 10 200 python tools/ab/ab_bits.py $L 1135 40000000 pl 2>&1 | tail -1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py -x -q -k "packed or fuzz or reference_order" 2>&1 | tail -3
