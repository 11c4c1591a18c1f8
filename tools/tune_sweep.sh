run() { echo -n "$1: "; env $1 timeout 200 python bench.py --steps 200 --no-fp32 --no-cpu-baseline --no-live-traffic --dqn-steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; }
for r in 1 2; do
run IVOSW_TUNE_STAGGER=0
run IVOSW_TUNE_STAGGER=60000
run IVOSW_TUNE_STAGGER=110000
run IVOSW_TUNE_NT=0
run IVOSW_TUNE_NT=1
run IVOSW_TUNE_NT=2
run IVOSW_TUNE_NT=3
done
