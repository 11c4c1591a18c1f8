#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the tower at evaluation batch sizes (the real two-stream path, not the per-layer report, which
# serialises the launches on one stream): per-kernel time per pass at B = 100 / 140 / 256, for the ranking VERDICT r5 item 4 asks for.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in ${@:-100 140 256}; do
  out=gpurun_out/evtrace_$B; rm -rf $out; mkdir -p $out
  timeout 240 rocprofv3 --kernel-trace --stats -d $out -o t -- python bench.py --tower-only --batch $B --steps 40 --warmup 10 > $out/log.txt 2>&1
  db=$(ls $out/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/evtrace_$B.txt
  rm -rf $out
done
