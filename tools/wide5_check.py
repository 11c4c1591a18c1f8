"""res5 identity blocks as ONE kernel per block (FUSE_WIDE5 = 1: two frames per workgroup, 2: one frame) against the three
layer kernels (0): stage output within bf16 ulps, scores within the bf16 tolerance.  Tuning aid, GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivos_w_amd import _lib as L, synth  # noqa: E402
from ivos_w_amd.models.assessment import AssessNet  # noqa: E402

dev = torch.device("cuda:0")
net = AssessNet(precision="bf16")
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.assessnet_state_dict(0).items()})
net.to(dev).eval()
lib = L.lib()
for B in (8, 3):
    tf, tp = synth.assess_inputs(B, seed=5 + B, structured=True)
    ttf, ttp = torch.from_numpy(tf).to(dev), torch.from_numpy(tp).to(dev)
    outs = {}
    for mode in (0, 1, 2):
        lib.ivosw_tune_set(b"FUSE_WIDE5", mode)
        s, t = net.forward_tap(ttf, ttp, "res5")
        outs[mode] = (s.float().cpu().numpy(), t.float().cpu().numpy())
    lib.ivosw_tune_set(b"FUSE_WIDE5", 0)
    for mode in (1, 2):
        a, b = outs[mode][1], outs[0][1]
        print(f"B={B} FUSE_WIDE5={mode}: max |d| / max = {np.abs(a - b).max() / np.abs(b).max():.3e}, scores rel {np.abs(outs[mode][0] - outs[0][0]).max() / np.abs(outs[0][0]).max():.3e}")
