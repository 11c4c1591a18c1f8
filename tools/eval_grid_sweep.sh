#!/bin/bash
# frames/s at evaluation-size batches with the persistent res2 chain kernel's grid at #CUs (default), #CUs / 2 and 3/4 (two streams from 64 units:
# two grid-of-#CUs launches cannot co-run, two of #CUs / 2 can); alternating, two rounds; run on the GPU box
run() { env $2 python bench.py --batch $1 --steps 200 --warmup 10 --no-cpu-baseline --no-live-traffic --no-clock-probe --no-fp32 --workload assess --dqn-steps 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=$1', '$2', d['value'])"; }
for r in 1 2; do for B in ${@:-100 140 200}; do
  run $B "X=0"
  run $B "IVOSW_TUNE_R2C_GRID=128"
  run $B "IVOSW_TUNE_R2C_GRID=192"
done; done
