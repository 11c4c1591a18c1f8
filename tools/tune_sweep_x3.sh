#!/bin/bash
# bf16x3 mode (batch 256) under one tunable changed at a time, defaults (X=0) interleaved, two rounds; run on the GPU box
run() { env $1 python bench.py --precision bf16x3 --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic --no-clock-probe --no-fp32 --workload assess --dqn-steps 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  for v in X=0 IVOSW_TUNE_F32_CHUNK=32 IVOSW_TUNE_F32_CHUNK=128 IVOSW_TUNE_NMAJOR=1 IVOSW_TUNE_NMAJOR=2 X=0 IVOSW_TUNE_LW=4 IVOSW_TUNE_NK=4 IVOSW_TUNE_NK=16 IVOSW_TUNE_STREAMS2_F32=0 IVOSW_TUNE_WS=0; do
    run $v
  done
done
