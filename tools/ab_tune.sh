for r in 1 2 3; do for v in 0 1; do
echo -n "FRONT_STAGGER=$v: "; IVOSW_TUNE_FRONT_STAGGER=$v timeout 200 python bench.py --steps 200 --no-fp32 --no-cpu-baseline --dqn-steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done; done
