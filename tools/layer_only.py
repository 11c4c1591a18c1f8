"""Per-layer HIP-event table of one-stream forward passes at B = 256 (bf16) with NO result check: for timing library variants whose
results are wrong by construction (ablation builds).  usage: python tools/layer_only.py out.txt [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivos_w_amd import _lib as L, synth  # noqa: E402
from ivos_w_amd.models.assessment import AssessNet  # noqa: E402

out = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
lib = L.lib()
net = AssessNet(precision="bf16")
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.assessnet_state_dict(0).items()})
net.to(dev).eval()
tf, tp = synth.assess_inputs(8, seed=1, structured=True)
tf = torch.from_numpy(tf).to(dev).repeat(B // 8, 1, 1, 1).contiguous()
tp = torch.from_numpy(tp).to(dev).repeat(B // 8, 1, 1).contiguous()
L.tune_set(b"STREAMS2", 0)
for _ in range(20):
    net(tf, tp)
torch.cuda.synchronize()
lib.ivosw_profile_start()
for _ in range(4):
    net(tf, tp)
buf = ctypes.create_string_buffer(1 << 16)
lib.ivosw_profile_report(buf, len(buf))
t2, c2 = ctypes.c_double(0), ctypes.c_int(0)
lib.ivosw_profile_stop(ctypes.byref(t2), ctypes.byref(c2))
open(out, "w").write(buf.value.decode())
print(buf.value.decode())
