#!/bin/bash
# variant_bench.sh "ENV=..." ROWPATTERN [rounds]: time each library variant under tools/_variants with the assess bench (alternating rounds);
# prints frames/s and the layer-report rows matching ROWPATTERN
envs=$1; pat=$2; rounds=${3:-2}
cp ivos-w_amd/libivosw_hip.so /tmp/lib_orig.so
for r in $(seq 1 $rounds); do for v in tools/_variants/lib_*.so; do
  cp $v ivos-w_amd/libivosw_hip.so
  env $envs python bench.py --batch 256 --steps 80 --warmup 5 --no-cpu-baseline --no-live-traffic --no-clock-probe --workload assess --layer-report /tmp/vb.layers > /tmp/vb.log 2>&1
  python - "$v" "$r" "$pat" <<'PY'
import json, sys, re
l=[x for x in open("/tmp/vb.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
rows=[x.split() for x in open("/tmp/vb.layers") if re.search(sys.argv[3], x)]
print(sys.argv[1].split("lib_")[-1], "round", sys.argv[2], d.get("value"), "frames/s |", " ; ".join(f"{r[7]}x{r[8]}us" for r in rows))
PY
done; done
cp /tmp/lib_orig.so ivos-w_amd/libivosw_hip.so
