"""A/B of the two res2 kernels (res2_chain.hip, RES2_CHAIN=1, against res2_stage.hip, RES2_CHAIN=0) on one box - GPU only.
usage: python tools/res2_chain_ab.py [B=256] [rounds=3]   (us per launch on one stream, alternating rounds; outputs compared)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivos_w_amd import _lib as L, synth  # noqa: E402
L.use_probe_lib()          # libivosw_probe.so: the product entries + the tuning probes (include/ivosw_probe.h)
from ivos_w_amd.models.assessment import AssessNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
lib = L.lib()
net = AssessNet(precision="bf16")
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.assessnet_state_dict(0).items()})
net.to(dev).eval()
packed = net._ensure_packed()
g = torch.Generator(device=dev).manual_seed(0)
x = torch.relu(torch.randn(B, 64, 64, 64, device=dev, generator=g)).to(torch.bfloat16)      # post-ReLU, post-pool statistics
st = L.stream_ptr(dev)
out = {}
for mode in (1, 0):
    out[mode] = (torch.zeros(B, 32, 32, 256, device=dev, dtype=torch.bfloat16), torch.zeros(B, 64, 64, 128, device=dev, dtype=torch.bfloat16))


def run(mode):
    y, t1 = out[mode]
    L.check(lib.ivosw_res2_stage_probe(L.dptr(packed), L.dptr(x), L.dptr(y), L.dptr(t1), B, 1, None, st), "probe")


def timed(mode, n=20):
    L.tune_set(b"RES2_CHAIN", mode)
    for _ in range(3):
        run(mode)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        run(mode)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


try:
    timed(1, 100)        # warm clocks
    for r in range(rounds):
        a, b = timed(1), timed(0)
        print(f"round {r}: B = {B}: chain {a:8.1f} us   stage {b:8.1f} us   ({100 * (a - b) / b:+.1f} %)   chain {2 * B * 1006632960 / a / 1e6:.0f} algorithmic TFLOP/s")
finally:
    L.tune_set(b"RES2_CHAIN", 1)
# phase timeline of the chain kernel (wave 0's stamps, mean over the workgroups)
ts = torch.zeros(B * 32, 8, device=dev, dtype=torch.int64)
L.tune_set(b"RES2_CHAIN", 1)
y, t1 = out[1]
for _ in range(3):
    L.check(lib.ivosw_res2_stage_probe(L.dptr(packed), L.dptr(x), L.dptr(y), L.dptr(t1), B, 1, L.dptr(ts), st), "probe")
torch.cuda.synchronize()
t = ts.cpu().numpy().astype(np.float64)[:, :7]
t = t[t[:, 6] > 0]          # (the persistent form stamps one tile per workgroup: one row per CU)
d = np.diff(t, axis=1).mean(axis=0)
names = ["prologue: halo + first group", "A0 (conv1 on 14 x 22)", "block 0", "block 1", "block 2 + res3 conv1", "stores"]
mf = [0, 3 * 9, 2 + 144 + 8 * 17 + 2 + 64, 2 + 144 + 8 * 13 + 2 + 64, 2 + 72 + 8 * 7 + 4 + 64, 0]        # MFMAs of the busiest wave
for n, c, m in zip(names, d, mf):
    print(f"  {n:32s} {c:8.0f} cycles   ({m * 32:6d} of MFMA issue)")
print(f"  {'total':32s} {(t[:, 6] - t[:, 0]).mean():8.0f} cycles   ({sum(mf) * 32:6d})")
for i, nm in enumerate(("y2 (even pixels)", "t1out")):
    a, b = out[1][i].float(), out[0][i].float()
    print(f"{nm}: max |chain - stage| = {(a - b).abs().max().item():.4f} of max {b.abs().max().item():.2f}; mean {(a - b).abs().mean().item():.2e} of mean {b.abs().mean().item():.3f}")
