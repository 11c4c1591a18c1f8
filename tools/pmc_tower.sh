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
fam = ("conv_igemm", "conv1x1_wide", "bneck", "res2_stage", "res2_chain_kernel", "gemm_8phase", "stage_first", "conv3x3_patch", "stem_pool", "roi_sample", "bbox_scan")
for k in sorted(agg):
    if not any(x in k for x in fam):
        continue
    print(k[:120])
    for c, (n, v) in sorted(agg[k].items()):
        print(f"    {c:34s} calls {n:4d}  per call {v / n:18.1f}")
print("""
# ---- derived shares (per kernel and launch; GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of the launch = GUI / 8; 256 CUs x 4 SIMDs;
# SQ_* wave counters in quad-cycles).  mfma = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); issue / wait_pipe = SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY over
# SQ_WAVE_CYCLES, parked = the rest (s_waitcnt / s_barrier); lds = SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs); conflict = share of the LDS cycles that are
# bank-conflict cycles; tcp = texture-cache accesses per CU-cycle (one 64-byte access per clock and CU at most).  rocprofv3 --pmc SERIALISES the dispatches:
# every launch here has the chip to itself, so a 128-workgroup launch (res4 chain: one frame per workgroup) shows half the share its CUs reach in the
# two-stream forward, where the other half's launch fills the remaining CUs.
# kernel                                                                 calls    Mcyc   mfma  issue wait_pipe  parked    lds conflict    tcp""")
def per(k, c):
    n, v = agg[k].get(c, [0, 0.0])
    return v / n if n else 0.0
for k in sorted(agg, key=lambda k: -per(k, "GRBM_GUI_ACTIVE") * agg[k].get("GRBM_GUI_ACTIVE", [0, 0])[0]):
    if not any(x in k for x in fam) or per(k, "GRBM_GUI_ACTIVE") == 0:
        continue
    cyc = per(k, "GRBM_GUI_ACTIVE") / 8.0
    wc = per(k, "SQ_WAVE_CYCLES") or 1.0
    issue, waitp = per(k, "SQ_ACTIVE_INST_ANY") / wc, per(k, "SQ_WAIT_INST_ANY") / wc
    lds = per(k, "SQ_LDS_IDX_ACTIVE")
    print(f"# {k[:70]:70s} {agg[k]['GRBM_GUI_ACTIVE'][0]:5d} {cyc / 1e6:7.3f} {per(k, 'SQ_VALU_MFMA_BUSY_CYCLES') / (cyc * 1024):6.3f} {issue:6.3f} {waitp:9.3f} "
          f"{max(0.0, 1 - issue - waitp):7.3f} {lds / (cyc * 256):6.3f} {(per(k, 'SQ_LDS_BANK_CONFLICT') / lds if lds else 0):8.3f} {per(k, 'TCP_TOTAL_CACHE_ACCESSES_sum') / (cyc * 256):6.3f}")
PY
