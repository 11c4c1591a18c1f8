#!/bin/bash
# res4 on half-frame tiles (HALF16_MAX = largest launch that uses them; 0 = never) against the frame / stage kernels, by batch size
for B in 32 64 100 128 160 200 256 272 300 340 384; do for v in 0 100000; do
echo -n "B=$B HALF16_MAX=$v: "; IVOSW_TUNE_HALF16_MAX=$v timeout 200 python bench.py --batch $B --steps 200 --no-fp32 --no-cpu-baseline --no-live-traffic --dqn-steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
