python -m pytest tests/test_gpu_assess.py -x -q 2>&1 | tail -5
python bench.py --steps 40 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline'].get('family_ms'))"
IVOSW_TUNE_RES2_CHAIN=0 python bench.py --steps 40 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])"
python bench.py --steps 40 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])"
IVOSW_TUNE_R2C_PERSIST=0 python bench.py --steps 40 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])"
