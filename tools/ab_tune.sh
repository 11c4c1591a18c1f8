#!/bin/bash
# A/B of one tunable on one box, alternating runs (boxes of the pool differ by +-3 %): tools/ab_tune.sh KEY v0 v1 [extra bench args]
key=$1; v0=$2; v1=$3; shift 3
for r in 1 2 3; do for v in $v0 $v1; do
echo -n "$key=$v: "; env IVOSW_TUNE_$key=$v timeout 200 python bench.py --steps 200 --no-fp32 --no-cpu-baseline --no-live-traffic --dqn-steps 5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d.get('front',{}).get('roi_us'), d.get('front',{}).get('bbox_us'))"
done; done
