#!/bin/bash
# for each library variant under tools/_variants: the res2 stage bit-identity test, then the phase probe (two alternating rounds)
cp ivos-w_amd/libivosw_hip.so /tmp/orig.so
for v in tools/_variants/lib_*.so; do cp $v ivos-w_amd/libivosw_hip.so; echo "== $v"; timeout 600 python -m pytest tests/test_gpu_assess.py -x -q -m gpu -k "res2_stage_kernel_is_bit_identical" 2>&1 | tail -2; done
for r in 1 2; do for v in tools/_variants/lib_*.so; do cp $v ivos-w_amd/libivosw_hip.so; echo "== $v round $r"; timeout 300 python tools/res2_stage_probe.py 256 2>&1 | grep -v amdgpu.ids; done; done
cp /tmp/orig.so ivos-w_amd/libivosw_hip.so
