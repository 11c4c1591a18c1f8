for r in 1 2 3; do for v in 1 2; do
echo -n "DF3=$v round $r: "; IVOSW_TUNE_DF3=$v python bench.py --steps 150 --warmup 10 --no-fp32 --no-cpu-baseline --no-live-traffic --no-clock-probe --dqn-steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done; done
