"""Experiment: one 256-frame forward on one stream vs two 128-frame forwards on two streams (tuning aid, GPU only)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivos_w_amd import synth  # noqa: E402
from ivos_w_amd.models.assessment import AssessNet  # noqa: E402

dev = torch.device("cuda:0")


def make():
    net = AssessNet(precision="bf16")
    sd = synth.assessnet_state_dict(0)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return net.to(dev).eval()


tf, tp = synth.assess_inputs(8, seed=3, structured=True)
tf = torch.from_numpy(np.repeat(tf, 32, 0)).to(dev)
tp = torch.from_numpy(np.repeat(tp, 32, 0)).to(dev)
n0 = make()
print("one stream x 256", end=": ")


def bench(fn, frames):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{dt * 1e3:.3f} ms, {frames / dt:.0f} frames/s")


bench(lambda: n0(tf, tp), 256)
for ns, per in ((2, 128), (4, 64), (2, 256), (4, 128), (3, 256), (8, 32)):
    nets = [make() for _ in range(ns)]
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    xs = [(tf[(i * per) % 256:(i * per) % 256 + per].contiguous(), tp[(i * per) % 256:(i * per) % 256 + per].contiguous()) for i in range(ns)]

    def run():
        out = []
        for n, st, (a, b) in zip(nets, streams, xs):
            with torch.cuda.stream(st):
                out.append(n(a, b))
        return out

    print(f"{ns} streams x {per}", end=": ")
    bench(run, ns * per)
    del nets
bench(lambda: n0(tf, tp), 256)
