for c in 8 16 32 64 128; do
echo -n "chunk=$c: "; timeout 300 python bench.py --precision fp32 --chunk $c --steps 6 --warmup 2 --min-warm-s 0.3 --no-cpu-baseline --no-live-traffic --dqn-steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launches_per_step'])"
done
