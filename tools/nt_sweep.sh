#!/bin/bash
# non-temporal stores / residual loads of the layer kernels (tunable NT: bit 0 stores, bit 1 residual loads) with the snake order, alternating runs
for r in 1 2 3; do for v in 3 0 2 1; do
echo -n "NT=$v round $r: "; IVOSW_TUNE_NT=$v python bench.py --steps 150 --warmup 10 --no-fp32 --no-cpu-baseline --no-live-traffic --no-clock-probe --dqn-steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])"
done; done
