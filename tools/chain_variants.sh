#!/bin/bash
# chain_variants.sh: A/B of compile-time variants of res2_chain.hip (ring depth, fragment lead, deferred piece issue) through
# tools/res2_chain_ab.py, which loads libivosw_probe.so.  Build part (no GPU): chain_variants.sh build; run part (GPU box): chain_variants.sh run
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/ivos-w_amd/csrc
mode=${1:-run}
declare -A V=( [base]="" [gf16]="-DRC_GF=16 -DRC_GROUPS_N=4" )
if [ "$mode" = build ]; then
  mkdir -p $root/tools/_variants
  for n in "${!V[@]}"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form ${V[$n]} -c $csrc/res2_chain.hip -o /tmp/rc_$n.o || exit 1
    objs=""
    for s in capi.cpp brain.hip dqn.hip assess_front.hip conv.hip bottleneck.hip bottleneck_wide.hip res2_stage.hip gemm_8phase.hip stage_first.hip stem.hip assess.hip metrics.hip seg_epilogue.hip p2p.hip; do
      if [ -f $csrc/build/$s.probe.o ]; then objs="$objs $csrc/build/$s.probe.o"; else objs="$objs $csrc/build/$s.o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/_variants/libprobe_$n.so $objs /tmp/rc_$n.o || exit 1
  done
  ls -la $root/tools/_variants/libprobe_*.so
else
  cp $root/ivos-w_amd/libivosw_probe.so /tmp/libprobe_orig.so
  for r in 1 2; do for n in base gf16; do
    cp $root/tools/_variants/libprobe_$n.so $root/ivos-w_amd/libivosw_probe.so
    echo "== $n round $r: $(python $root/tools/res2_chain_ab.py 256 1 2>&1 | grep -E 'round 0|total' | tr '\n' ' ')"
  done; done
  cp /tmp/libprobe_orig.so $root/ivos-w_amd/libivosw_probe.so
fi
