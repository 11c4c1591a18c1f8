"""Per-kernel sums of rocprofv3 --pmc counters (one pass) as a table; ratios against SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE
where present.  usage: pmc_sq_summary.py counter_collection.csv [min_calls]"""
import csv
import sys
from collections import defaultdict


def main():
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    names = sorted({c for v in agg.values() for c in v})
    print("# sums over all dispatches of a kernel; columns after the raw counters: MFMA busy cycles per CU-cycle (GRBM), LDS active / wave cycles")
    print(f"{'calls':>6} " + " ".join(f"{n[-22:]:>22}" for n in names) + "  kernel")
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k].get("GRBM_GUI_ACTIVE", 0))):
        v = agg[k]
        print(f"{len(calls[k]):6d} " + " ".join(f"{v.get(n, 0):22.4g}" for n in names) + f"  {k[:100]}")
        if "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0:
            g = v["GRBM_GUI_ACTIVE"]
            extra = []
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                extra.append(f"mfma_busy/(gui*256CU*4simd)={v['SQ_VALU_MFMA_BUSY_CYCLES'] / (g * 1024):.3f}")
            if "SQ_BUSY_CYCLES" in v:
                extra.append(f"sq_busy/gui={v['SQ_BUSY_CYCLES'] / g:.2f}")
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU_MFMA_MOPS_BF16"):
                if n in v and v.get("SQ_WAVE_CYCLES", 0) > 0:
                    extra.append(f"{n[3:].lower()}/wave_cyc={v[n] / v['SQ_WAVE_CYCLES']:.3f}")
            print("       " + "  ".join(extra))


if __name__ == "__main__":
    main()
