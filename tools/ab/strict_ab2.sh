for l in tools/ab/libstrict_nobuf.so snpmatch_amd/libsnpmatch_hip.so; do for extra in "" "--packed"; do
SNPMATCH_HIP_LIB=$l timeout -k 10 240 python bench.py --mode strict $extra --n-snp 6250000 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', '$extra', round(d['ms_per_step'],3), round(d['roofline']['avg_ms'],3), round(d['roofline']['frac'],3))"
done; done
