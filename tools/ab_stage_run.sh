for i in 1 2 3; do
for v in 1 0; do
IVOSW_TUNE_STAGE_RUN=$v timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --dqn-steps 5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('STAGE_RUN=$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done
done
