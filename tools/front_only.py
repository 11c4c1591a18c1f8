"""The assessment front end alone (mask -> box, ROI crop) on the bench's B = 256 inputs - a target for rocprofv3 passes (GPU only)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
args = argparse.Namespace(precision="bf16", chunk=0, batch=256)
net, tf, tp = bench.build_assess(args, 0, dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    print(bench.bench_front(args, dev, tf, tp))
