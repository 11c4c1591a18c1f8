"""Phase timeline of the wide fused bottleneck kernel (res4 identity block) — tuning aid, GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivos_w_amd import _lib as L  # noqa: E402
L.use_probe_lib()          # libivosw_probe.so: the product entries + the tuning probes (include/ivosw_probe.h)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Cm = int(sys.argv[2]) if len(sys.argv) > 2 else 256      # 256: res4 frame kernel, 128 / 64: halo kernels of res3 / res2
H, Cin = {256: 16, 128: 32, 64: 64}[Cm], 4 * Cm
NWG = B * {256: 1, 128: 4, 64: 32}[Cm]
dev = torch.device("cuda:0")
lib = L.lib()
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, H, H, Cin, device=dev, generator=g).to(torch.bfloat16)
y = torch.empty_like(x)
wa = (torch.randn(Cm, Cin, device=dev, generator=g) / Cin ** 0.5).to(torch.bfloat16)
wb = (torch.randn(Cm, 9 * Cm, device=dev, generator=g) / (9 * Cm) ** 0.5).to(torch.bfloat16)
wc = (torch.randn(Cin, Cm, device=dev, generator=g) / Cm ** 0.5).to(torch.bfloat16)
ba, bb, bc = (torch.randn(n, device=dev, generator=g) * 0.1 for n in (Cm, Cm, Cin))
frag = torch.zeros(2 * (Cm * Cin * 2 + 9 * Cm * Cm) + 256, device=dev, dtype=torch.uint8)
ts = torch.zeros(NWG, 8, device=dev, dtype=torch.int64)
st = L.stream_ptr(dev)


def run(tsbuf):
    L.check(lib.ivosw_bneck_wide_probe(L.dptr(x), L.dptr(y), L.dptr(wa), L.dptr(ba), L.dptr(wb), L.dptr(bb), L.dptr(wc), L.dptr(bc),
                                       L.dptr(frag), B, H, H, Cin, Cm, L.dptr(tsbuf) if tsbuf is not None else None, st), "probe")


for _ in range(3):
    run(None)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run(None)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
flops = 2.0 * B * H * H * (Cin * Cm + 9 * Cm * Cm + Cin * Cm)
print(f"B={B}: {us:.1f} us/launch incl. 3 fragpack launches, {flops / us / 1e6:.1f} TFLOP/s")
run(ts)
torch.cuda.synchronize()
t = ts.cpu().numpy().astype(np.float64)
d = np.diff(t[:, :7], axis=1)
names = ["A k-loop", "t1 store", "B taps", "t2 store", "C up to the last store pass", "C last store pass"]
for i, n in enumerate(names):
    print(f"{n:24s} {d[:, i].mean():9.0f} {np.percentile(d[:, i], 10):8.0f} {np.percentile(d[:, i], 90):8.0f}")
print(f"{'total':24s} {(t[:, 6] - t[:, 0]).mean():9.0f}   (shader clocks per workgroup)")
xf = x[:2].float().permute(0, 3, 1, 2)
t1 = torch.relu(torch.nn.functional.conv2d(xf, wa.float()[:, :, None, None], ba)).to(torch.bfloat16).float()
wb4 = wb.float().view(Cm, 3, 3, Cm).permute(0, 3, 1, 2)
t2 = torch.relu(torch.nn.functional.conv2d(t1, wb4, bb, padding=1)).to(torch.bfloat16).float()
ref = torch.relu(torch.nn.functional.conv2d(t2, wc.float()[:, :, None, None], bc) + xf).permute(0, 2, 3, 1)
print(f"max rel err vs torch reference: {(y[:2].float() - ref).abs().max().item() / ref.abs().max().item():.2e}")
