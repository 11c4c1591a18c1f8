#!/bin/bash
# Run ON THE GPU BOX (gpurun): kernel-trace + the two HBM-traffic PMC passes of the default bench command.
# Outputs under gpurun_out/prof_$1/ ; copy the summaries you want judged into profiles/.
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python bench.py --steps 4 --warmup 1 --min-warm-s 0 --no-fp32 --no-cpu-baseline --no-live-traffic --no-clock-probe --dqn-steps 20"
PASSES=5   # warmup 1 + timed 4 (the span timing runs inside the timed region; --no-fp32 keeps the fp32 conv_igemm launches out of the family)
# the trace run takes enough passes at warm clocks for its per-launch wall time to be comparable with the un-profiled line's avg_launch_us
TRACE_BENCH="python bench.py --steps 60 --warmup 10 --min-warm-s 1 --no-fp32 --no-cpu-baseline --no-live-traffic --no-clock-probe --dqn-steps 20"
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $TRACE_BENCH > $out/bench_trace.log 2>&1
db=$(ls $out/trace/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db > $out/kernel_trace_summary.txt
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o f --output-format csv -- $BENCH > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o w --output-format csv -- $BENCH > $out/bench_write.log 2>&1
python tools/pmc_summary.py $out/pmc_fetch/f_counter_collection.csv $out/pmc_write/w_counter_collection.csv "conv_igemm|conv1x1_wide|bneck|res2_stage|res2_chain_kernel|gemm_8phase|stage_first|conv3x3_patch|stem_pool" $PASSES $out/pmc_traffic.json > $out/pmc_hbm_traffic.txt
# the un-profiled line (never compare a profiled run with an un-profiled one)
python bench.py --steps 250 --warmup 10 --layer-report $out/layers.txt > $out/bench.json.log 2>&1
tail -1 $out/bench.json.log
head -12 $out/kernel_trace_summary.txt
tail -3 $out/pmc_hbm_traffic.txt
rm -rf $out/trace/*.db.tmp
ls -la $out $out/trace | head -30
