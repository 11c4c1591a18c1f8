cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r02f2; rm -rf $out; mkdir -p $out
TRACE_BENCH="python bench.py --steps 60 --warmup 10 --min-warm-s 1 --no-fp32 --no-cpu-baseline --no-live-traffic --dqn-steps 20"
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $TRACE_BENCH > $out/bench_trace.log 2>&1
db=$(ls $out/trace/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db > $out/kernel_trace_summary.txt
tail -1 $out/bench_trace.log | cut -c1-900
grep family $out/kernel_trace_summary.txt
rm -f $out/trace/*.db
