#!/bin/bash
# A/B of the res2 stage kernel (RES2_STAGE=1) against the per-block kernels (0): alternating bench runs + per-layer tables
mkdir -p gpurun_out
for r in 1 2; do
  for m in 1 0; do
    IVOSW_TUNE_RES2_STAGE=$m python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-live-traffic --workload assess \
        --layer-report gpurun_out/layers_stage$m.txt > gpurun_out/ab_stage${m}_r$r.log 2>&1
    python - <<PY
import json
l=[x for x in open("gpurun_out/ab_stage${m}_r$r.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("RES2_STAGE=$m round $r:", d.get("value"), "frames/s", d.get("ms_per_step"), "ms", "frac", d.get("roofline",{}).get("frac"))
PY
  done
done
echo "--- stage=1 layers"; cat gpurun_out/layers_stage1.txt; echo "--- stage=0 layers"; head -5 gpurun_out/layers_stage0.txt
