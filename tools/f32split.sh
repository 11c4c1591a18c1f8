for r in 1 2; do for v in 0 1; do
echo -n "STREAMS2_F32=$v: "; IVOSW_TUNE_STREAMS2_F32=$v timeout 300 python bench.py --precision fp32 --steps 10 --warmup 2 --min-warm-s 0.5 --no-cpu-baseline --no-live-traffic --dqn-steps 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['check'])"
done; done
