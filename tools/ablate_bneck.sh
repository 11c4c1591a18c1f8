#!/bin/bash
# per-layer HIP-event table for a set of fused-bottleneck ablation bits (tunable BDBG) — tuning aid
for d in "$@"; do
  IVOSW_TUNE_BDBG=$d python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dqn-steps 5 --layer-report gpurun_out/layers_bd$d.txt > gpurun_out/bench_bd$d.log 2>&1
  echo "BDBG=$d: $(grep ' 0  1   1' gpurun_out/layers_bd$d.txt | head -1)"
done
