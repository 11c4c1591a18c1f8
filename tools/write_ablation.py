"""what would it buy if the last blocks of res3 and res4 wrote y at the even pixels only (what a forwarded conv1 + a
stride-2 reader would need)?  ABLATION - results are wrong by construction.  Needs a library whose bottleneck_wide.hip and assess.hip were
built with -DIVOSW_ABLATION=1 (tools/build_variant.sh ys2abl "-DIVOSW_ABLATION=1" bottleneck_wide.hip assess.hip; then
tools/variant_run.sh "python tools/write_ablation.py").  Alternates YS2ABL = 0 / 1, two streams, batch 256, frames/s over 60 passes each."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivos_w_amd import _lib as L, synth  # noqa: E402
from ivos_w_amd.models.assessment import AssessNet  # noqa: E402

B = 256
dev = torch.device("cuda:0")
net = AssessNet(precision="bf16")
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.assessnet_state_dict(0).items()})
net.to(dev).eval()
tf, tp = synth.assess_inputs(8, seed=1, structured=True)
tf = torch.from_numpy(tf).to(dev).repeat(B // 8, 1, 1, 1).contiguous()
tp = torch.from_numpy(tp).to(dev).repeat(B // 8, 1, 1).contiguous()
for _ in range(60):
    net(tf, tp)
torch.cuda.synchronize()
for rnd in range(3):
    for v in (0, 1):
        L.tune_set(b"YS2ABL", v)
        for _ in range(10):
            net(tf, tp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            net(tf, tp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"round {rnd} YS2ABL={v}: {60 * B / dt:9.0f} frames/s  {dt / 60 * 1e3:.3f} ms per pass", flush=True)
