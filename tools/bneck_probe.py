"""Phase timeline of the fused bottleneck kernel (s_memtime stamps per workgroup) — tuning aid, GPU only.
usage: python tools/bneck_probe.py [B] [H] [Cin] [Cmid] [BDBG]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivos_w_amd import _lib as L  # noqa: E402
L.use_probe_lib()          # libivosw_probe.so: the product entries + the tuning probes (include/ivosw_probe.h)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
Cin = int(sys.argv[3]) if len(sys.argv) > 3 else 256
Cm = int(sys.argv[4]) if len(sys.argv) > 4 else 64
dbg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda:0")
lib = L.lib()
lib.ivosw_tune_set(b"BDBG", dbg)
if len(sys.argv) > 6:
    lib.ivosw_tune_set(b"STAGGER", int(sys.argv[6]))
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, H, H, Cin, device=dev, generator=g).to(torch.bfloat16)
y = torch.empty(B, H, H, 4 * Cm, device=dev, dtype=torch.bfloat16)
wa = (torch.randn(Cm, Cin, device=dev, generator=g) / Cin ** 0.5).to(torch.bfloat16)
wb = (torch.randn(Cm, 9 * Cm, device=dev, generator=g) / (9 * Cm) ** 0.5).to(torch.bfloat16)
wc = (torch.randn(4 * Cm, Cm, device=dev, generator=g) / Cm ** 0.5).to(torch.bfloat16)
ba, bb, bc = (torch.randn(n, device=dev, generator=g) * 0.1 for n in (Cm, Cm, 4 * Cm))
DS = Cin != 4 * Cm
wd = (torch.randn(4 * Cm, Cin, device=dev, generator=g) / Cin ** 0.5).to(torch.bfloat16) if DS else None
bd = torch.randn(4 * Cm, device=dev, generator=g) * 0.1 if DS else None
zeros = torch.zeros(256, device=dev, dtype=torch.uint8)
nwg = B * (H // 16) ** 2
ts = torch.zeros(nwg, 16, device=dev, dtype=torch.int64)
st = L.stream_ptr(dev)


def run(tsbuf):
    L.check(lib.ivosw_bneck_probe(L.dptr(x), L.dptr(y), L.dptr(wa), L.dptr(ba), L.dptr(wb), L.dptr(bb), L.dptr(wc), L.dptr(bc),
                                  L.dptr(wd) if DS else None, L.dptr(bd) if DS else None, L.dptr(zeros), B, H, H, Cin, Cm, L.dptr(tsbuf) if tsbuf is not None else None, st), "probe")


for _ in range(3):
    run(None)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run(None)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
flops = 2.0 * B * H * H * (Cin * Cm + 9 * Cm * Cm + 4 * Cm * Cm)
print(f"B={B} H={H} Cin={Cin} Cmid={Cm} dbg={dbg}: {us:.1f} us/launch, {flops / us / 1e6:.1f} TFLOP/s, "
      f"{B * H * H * (Cin + 4 * Cm) * 2 / us / 1e3:.0f} GB/s (x in + y out)")
run(ts)
torch.cuda.synchronize()
t = ts.cpu().numpy().astype(np.float64)
d = np.diff(t[:, :11], axis=1)
names = ["A tile0 landed", "A tile1 landed", "A tile2 landed", "A tile3 landed", "A last compute", "A epilogue+sync", "B taps",
         "B exchange+epi", "C half0 + mfma1", "C epilogue 1"]
print("phase            mean     p10     p90   (s_memtime ticks; 100 MHz => x24 shader cycles if constant clock)")
for i, n in enumerate(names):
    print(f"{n:16s} {d[:, i].mean():8.0f} {np.percentile(d[:, i], 10):7.0f} {np.percentile(d[:, i], 90):7.0f}")
print("K-tile 0: landed->computed %.0f, ->barrier+issue T3 %.0f | K-tile 1: landed->computed %.0f, ->barrier+issue Wb %.0f" % (
    (t[:, 11] - t[:, 1]).mean(), (t[:, 12] - t[:, 11]).mean(), (t[:, 13] - t[:, 2]).mean(), (t[:, 14] - t[:, 13]).mean()))
tot = t[:, 10] - t[:, 0]
print(f"{'total':14s} {tot.mean():8.0f} {np.percentile(tot, 10):7.0f} {np.percentile(tot, 90):7.0f}")
span = t[:, 10].max() - t[:, 0].min()
print(f"kernel span {span:.0f} ticks; sum of WG time / span = {tot.sum() / span:.1f} concurrent WGs")
# check vs a torch reference on a few frames
xf = x[:2].float().permute(0, 3, 1, 2)
t1 = torch.relu(torch.nn.functional.conv2d(xf, wa.float()[:, :, None, None], ba)).to(torch.bfloat16).float()
wb4 = wb.float().view(Cm, 3, 3, Cm).permute(0, 3, 1, 2)
t2 = torch.relu(torch.nn.functional.conv2d(t1, wb4, bb, padding=1)).to(torch.bfloat16).float()
idt = torch.nn.functional.conv2d(xf, wd.float()[:, :, None, None], bd) if DS else xf
ref = torch.relu(torch.nn.functional.conv2d(t2, wc.float()[:, :, None, None], bc) + idt).permute(0, 2, 3, 1)
if dbg == 0:
    err = (y[:2].float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"max rel err vs torch fp32-accumulate reference (bf16 intermediates): {err:.2e}")
