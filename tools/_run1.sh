IVOSW_BENCH_DEBUG=1 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --no-fp32 --no-clock-probe 2>&1 | grep -v "^{" | tail -5
