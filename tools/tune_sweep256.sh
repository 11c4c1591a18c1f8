#!/bin/bash
# the headline (batch 256, bf16) under one tunable changed at a time, alternating with the default, two rounds; run on the GPU box
run() { env $1 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-live-traffic --no-clock-probe --no-fp32 --workload assess --dqn-steps 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], d['roofline']['frac'])"; }
for r in 1 2; do
  for v in X=0 IVOSW_TUNE_STAGGER=0 IVOSW_TUNE_STAGGER=60000 IVOSW_TUNE_STAGGER=160000 X=0 IVOSW_TUNE_DF3=1 IVOSW_TUNE_DF3=0 IVOSW_TUNE_DF3=4 IVOSW_TUNE_SNAKE=0 X=0 IVOSW_TUNE_G8=1 IVOSW_TUNE_NT=0 IVOSW_TUNE_NT=1 IVOSW_TUNE_FUSE_WIDE5=1 X=0 IVOSW_TUNE_FWD2=0 IVOSW_TUNE_FIRST3=0 IVOSW_TUNE_INPLACE4=0 IVOSW_TUNE_HALO_WLDS=0; do
    run $v
  done
done
