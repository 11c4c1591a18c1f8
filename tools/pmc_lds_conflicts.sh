#!/bin/bash
# LDS bank-conflict share per tower kernel (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE), one rocprofv3 --pmc pass of bench.py --tower-only.
# usage (on the GPU box): tools/pmc_lds_conflicts.sh [ENV=VALUE ...]     e.g. IVOSW_TUNE_PATCH_KEYXY=0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_lds; rm -rf $out; mkdir -p $out
env "$@" timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/p -o p --output-format csv -- python bench.py --tower-only --steps 1 --warmup 1 > $out/log.txt 2>&1 || echo "pmc pass failed"
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]].add(r["Dispatch_Id"])
print(f"{'calls':>5} {'lds cycles/call':>16} {'conflict share':>15}  kernel")
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_LDS_IDX_ACTIVE", 0)):
    a, c = agg[k].get("SQ_LDS_IDX_ACTIVE", 0), agg[k].get("SQ_LDS_BANK_CONFLICT", 0)
    if a > 1e6:
        print(f"{len(n[k]):5d} {a / len(n[k]):16.0f} {c / a:15.3f}  {k[:110]}")
PY
