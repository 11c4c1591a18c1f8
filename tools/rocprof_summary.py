"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as text: per-kernel totals and, for one
kernel family, a per-launch-shape breakdown.  usage: rocprof_summary.py results.db [family-substring]"""
import sqlite3
import subprocess
import sys


def demangle(n):
    n = n[:-3] if n.endswith(".kd") else n
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n


def main():
    db, fam = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "conv_igemm|conv1x1_wide|bneck|res2_stage|res2_chain_kernel|gemm_8phase|stage_first|conv3x3_patch|stem_pool")
    like = " or ".join(f"s.kernel_name like '%{x}%'" for x in fam.split("|"))
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc"))
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {db}\n# total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7}  kernel")
    for r in rows:
        print(f"{r[1]:7d} {r[2]/1e6:10.3f} {r[3]/1e3:9.2f} {r[4]/1e3:9.2f} {r[5]/1e3:9.2f} {100*r[2]/total:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d}  {demangle(r[0])[:150]}")
    print(f"\n# launches of *{fam}* by (instance, grid): one line per distinct layer shape class")
    rows = list(cur.execute(
        "select s.kernel_name, d.grid_size_x, count(*), sum(d.end-d.start), avg(d.end-d.start) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        f"where {like} group by s.kernel_name, d.grid_size_x order by 4 desc"))
    ft = sum(r[3] for r in rows)
    n = sum(r[2] for r in rows)
    print(f"# family total {ft/1e6:.3f} ms, {n} launches, avg {ft/max(n,1)/1e3:.2f} us")
    # With the two-stream split two launches of the family run side by side, so the sum of their durations counts that
    # time twice: what bench.py's roofline uses (one HIP-event span per forward pass, latest end - earliest start over
    # the two streams) is the WALL time during which a family kernel was running = the union of the dispatch intervals.
    iv = sorted(cur.execute(
        "select d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        f"where {like}"))
    wall, cs, ce = 0, None, None
    for a, b in iv:
        if cs is None or a > ce:
            if cs is not None:
                wall += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    if cs is not None:
        wall += ce - cs
    print(f"# family wall time (union of the {n} dispatch intervals) {wall/1e6:.3f} ms = {wall/max(n,1)/1e3:.2f} us per launch "
          f"(compare bench.py roofline.avg_launch_us; overlap factor {ft/max(wall,1):.2f})\n")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'pct':>6} {'blocks':>8}  instance")
    for r in rows:
        print(f"{r[2]:7d} {r[3]/1e6:10.3f} {r[4]/1e3:9.2f} {100*r[3]/ft:6.2f} {r[1]//256:8d}  {demangle(r[0])[:120]}")


if __name__ == "__main__":
    main()
