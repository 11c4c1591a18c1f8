#!/bin/bash
# stage kernel (RES2_STAGE=1) vs per-block kernels (0) at evaluation-size batches, alternating runs
for B in 32 64 100 160 256; do
  for r in 1 2; do
    for m in 1 0; do
      IVOSW_TUNE_RES2_STAGE=$m python bench.py --batch $B --steps 80 --warmup 5 --no-cpu-baseline --no-live-traffic --workload assess > gpurun_out/abb_${B}_${m}_$r.log 2>&1
      python - <<PY
import json
l=[x for x in open("gpurun_out/abb_${B}_${m}_$r.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("B=$B RES2_STAGE=$m round $r:", d.get("value"), "frames/s", d.get("ms_per_step"), "ms")
PY
    done
  done
done
