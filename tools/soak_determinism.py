"""Soak (GPU only): N back-to-back bf16 forwards at a few batch sizes, every result compared bit for bit with the first pass of its
size.  usage: soak_determinism.py [passes_per_size]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ivos_w_amd import synth  # noqa: E402
from test_gpu_assess import _variants, make_net  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
net = make_net(dev, "bf16")
tf, tp = synth.assess_inputs(16, seed=77, structured=True)
for B in (256, 100, 33):
    ttf, ttp = _variants(torch.from_numpy(tf).to(dev), torch.from_numpy(tp).to(dev), B)
    first = net(ttf, ttp).reshape(-1).clone()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(n):
        s = net(ttf, ttp).reshape(-1)
        bad += (s.view(torch.int32) != first.view(torch.int32)).sum()
    print(f"B={B}: {n} passes, {int(bad.item())} differing scores")
