"""Print the kernel timeline of one DQN step from a rocprofv3 --kernel-trace database (start, duration, stream/queue)."""
import glob
import sqlite3
import sys

db = sys.argv[1] if len(sys.argv) > 1 else glob.glob("gpurun_out/prof_dqn/*.db")[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
namecol = "kernel_name" if "kernel_name" in cols else "display_name"
rows = c.execute(f"select k.{namecol}, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.grid_size_y, d.grid_size_z, d.queue_id, d.stream_id "
                 f"from {kd} d join {ks} k on d.kernel_id = k.id order by d.start").fetchall()
# a step starts at its first kernel: the encoder launch when the draw + gather are folded into it (ivosw_dqn_step_drawn, round 4),
# else the gather kernel; the step printed is one from the MIDDLE of the run (the tail of a bench run is other legs)
fold = [i for i, r in enumerate(rows) if "clamp_adam_dev_reduce" in r[0]]
if fold:
    idx = [i for i, r in enumerate(rows) if "enc_fused_kernel" in r[0] and i < fold[-1]]
    a = idx[len(idx) // 2]
    b = next(i for i in idx if i > a)
else:
    idx = [i for i, r in enumerate(rows) if "replay_gather" in r[0] or "replay_draw_gather" in r[0]]
    a, b = idx[-3], idx[-2]
t0 = rows[a][1]
busy_end = t0
for r in rows[a:b]:
    name = r[0].split("(")[0].replace("ivosw::", "").replace("void ", "")[:34]
    print(f"{(r[1] - t0) / 1e3:8.1f} .. {(r[2] - t0) / 1e3:8.1f}  dur {(r[2] - r[1]) / 1e3:6.1f}  q{r[7]} s{r[8]}  grid {r[3] // r[4]}x{r[5]}x{r[6]}  {name}")
print("step span", (rows[b][1] - t0) / 1e3, "us; sum of durations", sum(r[2] - r[1] for r in rows[a:b]) / 1e3)
