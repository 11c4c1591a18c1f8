#!/bin/bash
# snake order of consecutive tower launches (tunable SNAKE) on / off, alternating runs: frames/s, tower ms per pass, res3 identity block us
for r in 1 2 3; do for v in 0 1; do
echo -n "SNAKE=$v round $r: "; IVOSW_TUNE_SNAKE=$v python bench.py --steps 150 --warmup 10 --no-fp32 --no-cpu-baseline --no-live-traffic --no-clock-probe --dqn-steps 5 --layer-report /tmp/rev.layers 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], end=' ')"; grep " 32   512   512  0" /tmp/rev.layers | awk '{print $9}'
done; done
