#!/bin/bash
# PMC passes (one counter group each, own timeout: an unsupported group makes rocprofv3 abort and hang) over tools/front_only.py - run on the GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout -k 5 120 rocprofv3 --pmc $c -d gpurun_out/pmc_front/$tag -o p --output-format csv -- python tools/front_only.py 1 > gpurun_out/pmc_front/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_front/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:40], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for (k, c), (n, v) in sorted(agg.items()):
        if "roi_sample" in k or "bbox_scan" in k:
            print(f"{k:42s} {c:32s} calls {n:4d}  per call {v / n:16.1f}")
PY
