"""HBM traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the
same command).  Units/corrections per MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of wide coalesced streaming reads, so reads are doubled; WRITE_SIZE is taken as is (uncalibrated).
usage: pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv [family[|family2]] [forward passes] [out.json]
With out.json: writes {"bytes_per_launch", "launches_per_pass", "bytes_per_pass", "source"} for bench.py's roofline.traffic."""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    fams = (sys.argv[3] if len(sys.argv) > 3 else "conv_igemm|conv1x1_wide|bneck|res2_stage|res2_chain_kernel|gemm_8phase|stage_first|conv3x3_patch|stem_pool").split("|")
    fam = "|".join(fams)
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    print("# HBM traffic from rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE as reported), KiB -> bytes")
    print(f"{'calls':>7} {'read_GB':>9} {'write_GB':>9} {'MB/launch':>10}  kernel")
    tot = [0, 0.0, 0.0]
    for k in sorted(f, key=lambda k: -f[k][1]):
        rd, wr = 2 * f[k][1] * 1024, w.get(k, [0, 0.0])[1] * 1024
        print(f"{f[k][0]:7d} {rd/1e9:9.3f} {wr/1e9:9.3f} {(rd+wr)/max(f[k][0],1)/1e6:10.2f}  {k[:110]}")
        if any(x in k for x in fams):
            tot[0] += f[k][0]; tot[1] += rd; tot[2] += wr
    print(f"\n# family *{fam}*: {tot[0]} launches, read {tot[1]/1e9:.3f} GB, write {tot[2]/1e9:.3f} GB, "
          f"{(tot[1]+tot[2])/max(tot[0],1)/1e6:.2f} MB per launch, {(tot[1]+tot[2])/steps/1e9:.3f} GB per forward pass ({steps} passes profiled)")
    if len(sys.argv) > 5:
        json.dump({"bytes_per_launch": (tot[1] + tot[2]) / max(tot[0], 1), "launches_per_pass": tot[0] / steps,
                   "bytes_per_pass": (tot[1] + tot[2]) / steps, "read_bytes_per_pass": tot[1] / steps, "write_bytes_per_pass": tot[2] / steps,
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; FETCH_SIZE x2 gfx950 correction, KiB units) over "
                             f"{steps} forward passes of `python bench.py`, kernels matching {fam}"}, open(sys.argv[5], "w"), indent=1)


if __name__ == "__main__":
    main()
