#!/bin/bash
# variant_run.sh "<command>" [rounds]: run a command under each library variant in tools/_variants (alternating rounds)
cmd=$1; rounds=${2:-1}
cp ivos-w_amd/libivosw_hip.so /tmp/lib_orig.so
for r in $(seq 1 $rounds); do for v in tools/_variants/lib_*.so; do
  cp $v ivos-w_amd/libivosw_hip.so
  echo "== $(basename $v) round $r"
  bash -c "$cmd"
done; done
cp /tmp/lib_orig.so ivos-w_amd/libivosw_hip.so
