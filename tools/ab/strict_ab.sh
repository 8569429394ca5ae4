mkdir -p gpurun_out/r02
for extra in "" "--packed"; do
timeout -k 10 240 python bench.py --mode strict $extra --n-snp 6250000 --steps 5 --warmup 2 --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$extra', d['ms_per_step'], d['value'], d['roofline']['frac'])"
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py -x -q 2>&1 | tail -5
