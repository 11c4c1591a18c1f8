"""Phase timeline of the res2 stage kernel (res2_stage.hip) - tuning aid, GPU only.
usage: python tools/res2_stage_probe.py [B=256]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivos_w_amd import _lib as L, synth  # noqa: E402
L.use_probe_lib()          # libivosw_probe.so: the product entries + the tuning probes (include/ivosw_probe.h)
from ivos_w_amd.models.assessment import AssessNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
lib = L.lib()
net = AssessNet(precision="bf16")
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.assessnet_state_dict(0).items()})
net.to(dev).eval()
packed = net._ensure_packed()
g = torch.Generator(device=dev).manual_seed(0)
x = torch.relu(torch.randn(B, 64, 64, 64, device=dev, generator=g)).to(torch.bfloat16)      # post-ReLU, post-pool statistics
y = torch.empty(B, 32, 32, 256, device=dev, dtype=torch.bfloat16)
t1 = torch.empty(B, 64, 64, 128, device=dev, dtype=torch.bfloat16)
ts = torch.zeros(B * 32, 16, device=dev, dtype=torch.int64)
st = L.stream_ptr(dev)


def run(tsbuf):
    L.check(lib.ivosw_res2_stage_probe(L.dptr(packed), L.dptr(x), L.dptr(y), L.dptr(t1), B, 1, L.dptr(tsbuf) if tsbuf is not None else None, st), "probe")


names = ["p DMA wait", "A0", "B0 (3x3 + exchange)", "C0/D0 (2 K halves)", "t1_1 store", "B1", "C1", "D1 + t1_2 store", "B2", "C2",
         "y2 out + D2"]
mf = [0, 12 * 2, 72 * 2, (32 + 16) * 2 * 2, 0, 54 * 2, 24 * 2, 32 * 2, 36 * 2, 16 * 2, 32 * 2]      # MFMAs per SIMD (two waves) on the critical path
modes = [0]
if lib.ivosw_ablation_build():
    modes = [0, 1, 2, 4, 3, 5, 6, 7]        # bits: 1 no weight loads, 2 pixel fragments read once per phase, 4 no MFMAs
cols = {}
for mode in modes:
    lib.ivosw_tune_set(b"R2DBG", mode)
    for _ in range(5):
        run(None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    N = 20
    for _ in range(N):
        run(None)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / N
    ts.zero_()
    run(ts)
    torch.cuda.synchronize()
    t = ts.cpu().numpy().astype(np.float64)
    d = np.diff(t[:, :12], axis=1)
    tot = (t[:, 11] - t[:, 0]).mean()
    cols[mode] = (us, d.mean(axis=0), tot)
    print(f"R2DBG={mode}: B={B}: {us:.1f} us/launch  {2 * B * 1006632960 / us / 1e6:.1f} algorithmic TFLOP/s; {tot:.0f} clocks per workgroup -> {tot * 32 * B / 256 / us / 1e3:.2f} GHz")
lib.ivosw_tune_set(b"R2DBG", 0)
print(f"{'phase':24s} {'MFMA':>6s} " + " ".join(f"{'dbg' + str(m):>7s}" for m in modes))
for i, n in enumerate(names):
    print(f"{n:24s} {mf[i] * 32:6d} " + " ".join(f"{cols[m][1][i]:7.0f}" for m in modes))
print(f"{'total':24s} {2912 * 8:6d} " + " ".join(f"{cols[m][2]:7.0f}" for m in modes))
print(f"{'us / launch':24s} {'':6s} " + " ".join(f"{cols[m][0]:7.0f}" for m in modes))
