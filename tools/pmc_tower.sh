#!/bin/bash
# PMC passes over the tower's kernels (bench.py --tower-only, 2 forward passes): one counter group per pass with its own timeout
# (an unsupported group makes rocprofv3 abort and hang).  Run on the GPU box; summary on stdout.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_tower; rm -rf $out; mkdir -p $out
for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-48)
  timeout -k 5 150 rocprofv3 --pmc $c -d $out/$tag -o p --output-format csv -- python bench.py --tower-only --steps 1 --warmup 1 > $out/$tag.log 2>&1 || echo "pass '$c' failed"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("gpurun_out/pmc_tower/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
fam = ("conv_igemm", "conv1x1_wide", "bneck", "res2_stage", "stage_first", "conv3x3_patch", "stem_pool", "roi_sample", "bbox_scan")
for k in sorted(agg):
    if not any(x in k for x in fam):
        continue
    print(k[:120])
    for c, (n, v) in sorted(agg[k].items()):
        print(f"    {c:34s} calls {n:4d}  per call {v / n:18.1f}")
PY
