#!/bin/bash
# frames/s at evaluation-size batches under a few threshold settings (alternating, two rounds); run on the GPU box
run() { env $2 python bench.py --batch $1 --steps 200 --warmup 10 --no-cpu-baseline --no-live-traffic --no-clock-probe --no-fp32 --workload assess --dqn-steps 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=$1', '$2', d['value'])"; }
for r in 1 2; do for B in 100 140 200 300; do
  run $B "X=0"
  run $B "IVOSW_TUNE_HALF16_MAX=128"
  run $B "IVOSW_TUNE_HALF16_MAX=64"
  run $B "IVOSW_TUNE_SMALL_GRID=192"
  run $B "IVOSW_TUNE_SNAKE_MIN=32"
  run $B "IVOSW_TUNE_DF3_MIN=48"
done; done
