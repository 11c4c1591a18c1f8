"""Is the DQN step host-bound?  Time the enqueue loop alone (no sync) against the total (tuning aid, GPU only)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = argparse.Namespace(replay=50000, minibatch=128)
dev = torch.device("cuda:0")
agent, replay, gen = bench.build_dqn(args, 0, dev)


def step():
    idx = torch.randint(0, len(replay), (128,), device=dev, generator=gen)
    agent.loss_and_grads(replay.sample(idx))
    agent.optimizer.step()


for _ in range(30):
    step()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / n:.1f} us/step, total {1e6 * (t2 - t0) / n:.1f} us/step (GPU drains {1e3 * (t2 - t1):.2f} ms after the loop)")
