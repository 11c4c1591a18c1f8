#!/bin/bash
# time each library variant under tools/_variants with the res2 stage probe (two rounds, alternating)
for r in 1 2; do for v in tools/_variants/lib_*.so; do cp $v ivos-w_amd/libivosw_hip.so; echo "== $v round $r"; python tools/res2_stage_probe.py 256 2>&1 | grep "R2DBG=0\|^total"; done; done
