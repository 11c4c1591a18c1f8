for B in 64 100 160 300; do for v in 16 1000; do
echo -n "B=$B STREAMS2_MIN=$v: "; IVOSW_TUNE_STREAMS2_MIN=$v timeout 200 python bench.py --batch $B --steps 200 --no-fp32 --no-cpu-baseline --no-live-traffic --dqn-steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
