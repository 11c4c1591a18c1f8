#!/bin/bash
# Run ON THE GPU BOX: the res2 stage kernel variant (RES2_STAGE=1) of the default bench command — kernel trace + the two HBM-traffic PMC passes,
# for the comparison with the per-block default in profiles/ (same box, same command as tools/profile_round.sh).
tag=${1:-r03_stage}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export IVOSW_TUNE_RES2_STAGE=1
BENCH="python bench.py --steps 4 --warmup 1 --min-warm-s 0 --no-fp32 --no-cpu-baseline --no-live-traffic --no-clock-probe --dqn-steps 20"
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python bench.py --steps 60 --warmup 10 --min-warm-s 1 --no-fp32 --no-cpu-baseline --no-live-traffic --no-clock-probe --dqn-steps 20 > $out/bench_trace.log 2>&1
db=$(ls $out/trace/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db > $out/kernel_trace_summary.txt
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o f --output-format csv -- $BENCH > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o w --output-format csv -- $BENCH > $out/bench_write.log 2>&1
python tools/pmc_summary.py $out/pmc_fetch/f_counter_collection.csv $out/pmc_write/w_counter_collection.csv "conv_igemm|conv1x1_wide|bneck|res2_stage|res2_chain_kernel|gemm_8phase|stage_first|conv3x3_patch|stem_pool" 5 $out/pmc_traffic.json > $out/pmc_hbm_traffic.txt
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum -d $out/pmc_tcp -o c --output-format csv -- $BENCH > $out/bench_tcp.log 2>&1
python - <<PY > $out/pmc_tcp_accesses.txt
import csv, collections
rows = list(csv.DictReader(open("$out/pmc_tcp/c_counter_collection.csv")))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    if any(s in k for s in ("res2_stage", "bneck_halo64s")):
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in agg.items():
    print(f"{k:60s} launches {n:4d}  TCP_TOTAL_CACHE_ACCESSES per launch {v / n / 1e6:8.1f} M")
PY
python bench.py --steps 120 --warmup 10 --no-cpu-baseline --layer-report $out/layers.txt > $out/bench.json.log 2>&1
tail -c 600 $out/bench.json.log; head -14 $out/kernel_trace_summary.txt; tail -3 $out/pmc_hbm_traffic.txt; cat $out/pmc_tcp_accesses.txt
rm -rf $out/trace/*.db.tmp $out/trace/*.db
