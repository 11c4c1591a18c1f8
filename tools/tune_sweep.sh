#!/bin/bash
# Sweep of stale-default candidates on one box (two rounds, alternating): tools/tune_sweep.sh "KEY=v" "KEY=v" ...
run() { echo -n "$1: "; env IVOSW_TUNE_$1 timeout 200 python bench.py --steps 200 --no-fp32 --no-cpu-baseline --no-live-traffic --dqn-steps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; }
for r in 1 2; do for kv in "$@"; do run $kv; done; done
