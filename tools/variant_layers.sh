#!/bin/bash
# variant_layers.sh ROWPATTERN [rounds]: layer-table rows of each library variant under tools/_variants (alternating rounds; results of ablation
# variants are WRONG by construction: the bench's own check is bypassed with --tower-only style timing via --layer-report only)
pat=$1; rounds=${2:-2}
cp ivos-w_amd/libivosw_hip.so /tmp/orig.so
for r in $(seq 1 $rounds); do for v in tools/_variants/lib_*.so; do cp $v ivos-w_amd/libivosw_hip.so
  python tools/layer_only.py /tmp/l.txt > /dev/null 2>&1
  echo "$(basename $v) round $r: $(grep -E "$pat" /tmp/l.txt | awk '{printf "%s/%s/%s:%s  ", $3,$4,$7,$9}')"; done; done
cp /tmp/orig.so ivos-w_amd/libivosw_hip.so
