"""Shader clock while the Double-DQN loop runs (GPU only): the chain uses a fraction of the chip, does the power manager keep the
clock up?  usage: dqn_clock_probe.py [spin_us]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ivos_w_amd import _lib as L  # noqa: E402
from ivos_w_amd.models.agent import AutoDqnLoop  # noqa: E402

spin = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
args = type("A", (), dict(replay=50000, minibatch=128))()
agent, replay, gen = bench.build_dqn(args, 0, dev)
auto = AutoDqnLoop(agent, replay, 128, draw_seed=2019, block=8)
auto.choice = "plain"
auto.run(256)
torch.cuda.synchronize(dev)
lib = L.lib()
side = torch.cuda.Stream(device=dev)
out = torch.zeros(16, 2, dtype=torch.int64, device=dev)
for i in range(16):
    L.check(lib.ivosw_clock_probe(out[i].data_ptr(), spin, ctypes.c_void_p(side.cuda_stream)), "clock_probe")
    auto.run(16)
torch.cuda.synchronize(dev)
o = out.cpu().numpy().astype(np.float64)
mhz = o[:, 0] / np.maximum(o[:, 1], 1.0) * 100.0
print("sclk under the DQN loop (MHz):", np.round(mhz, 0))
print("power W:", bench._read_power_w())
# idle clock for comparison
out.zero_()
for i in range(4):
    L.check(lib.ivosw_clock_probe(out[i].data_ptr(), spin, ctypes.c_void_p(side.cuda_stream)), "clock_probe")
    torch.cuda.synchronize(dev)
o = out.cpu().numpy().astype(np.float64)[:4]
print("sclk idle (probe alone):", np.round(o[:, 0] / np.maximum(o[:, 1], 1.0) * 100.0, 0))
