#!/bin/bash
for r in 1 2 3; do for v in 0 1; do
echo -n "INPLACE4=$v round $r: "; IVOSW_TUNE_INPLACE4=$v python bench.py --steps 150 --warmup 10 --no-fp32 --no-cpu-baseline --no-live-traffic --no-clock-probe --dqn-steps 5 --layer-report /tmp/ip.layers 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], end=' ')"; grep "1280   16  1024  1024  0" /tmp/ip.layers | awk '{print $9}'
done; done
