#!/bin/bash
# for each library variant under tools/_variants: the fused-first-block bit-identity test, then the layer-table row of stage_first_kernel and frames/s (alternating rounds)
cp ivos-w_amd/libivosw_hip.so /tmp/orig.so
for v in tools/_variants/lib_*.so; do cp $v ivos-w_amd/libivosw_hip.so; echo "== $v"; timeout 600 python -m pytest tests/test_gpu_assess.py -x -q -m gpu -k "fused_first_block" 2>&1 | tail -2; done
for r in 1 2 3; do for v in tools/_variants/lib_*.so; do cp $v ivos-w_amd/libivosw_hip.so
  python bench.py --batch 256 --steps 80 --warmup 5 --no-cpu-baseline --no-live-traffic --no-clock-probe --no-fp32 --workload assess --dqn-steps 20 --layer-report /tmp/l.txt 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v round $r', d['value'], d['roofline']['frac'])"; grep -E " -3 " /tmp/l.txt; done; done
cp /tmp/orig.so ivos-w_amd/libivosw_hip.so
