#!/bin/bash
# A/B of one tunable on the GPU box, alternating runs: ab_tune_batches.sh KEY "B1 B2 ..." [rounds]   (values 1 vs 0)
key=$1; batches=${2:-256}; rounds=${3:-2}
mkdir -p gpurun_out
for B in $batches; do
  for r in $(seq 1 $rounds); do
    for m in 1 0; do
      env IVOSW_TUNE_$key=$m python bench.py --batch $B --steps 80 --warmup 5 --no-cpu-baseline --no-live-traffic --no-clock-probe --workload assess --layer-report gpurun_out/ab_${key}_${B}_${m}.layers > gpurun_out/ab_${key}_${B}_${m}_$r.log 2>&1
      python - <<PY
import json
l=[x for x in open("gpurun_out/ab_${key}_${B}_${m}_$r.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("B=$B $key=$m round $r:", d.get("value"), "frames/s", d.get("ms_per_step"), "ms")
PY
    done
  done
done
