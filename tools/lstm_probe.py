"""Phase timeline of the MFMA forward recurrence (lstm_fwd_mfma_kernel): s_memtime stamps of step T/2 — tuning aid, GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivos_w_amd import _lib as L  # noqa: E402
L.use_probe_lib()          # libivosw_probe.so: the product entries + the tuning probes (include/ivosw_probe.h)
from ivos_w_amd import synth  # noqa: E402
from ivos_w_amd.models.agent import Brain  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = torch.device("cuda:0")
net = Brain().to(dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.brain_state_dict(0).items()})
x = torch.from_numpy(synth.brain_inputs(N, T, 3).astype(np.float32)).to(dev)
nwg = (2 * N + 3) // 4
ts = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
for _ in range(3):
    net(x)
L.lib().ivosw_lstm_probe(L.dptr(ts), None)
net(x)
torch.cuda.synchronize()
L.lib().ivosw_lstm_probe(None, None)
t = ts.cpu().numpy().astype(np.float64)
d = np.diff(t[:, :4], axis=1)
print(f"N={N} T={T}: {nwg} workgroups; ticks of step {T // 2} (mean / min / max over workgroups)")
for name, col in (("h reads + 128 MFMAs", 0), ("activations + state update + stores", 1), ("barrier", 2)):
    print(f"  {name:38s} {d[:, col].mean():8.0f} {d[:, col].min():8.0f} {d[:, col].max():8.0f}")
print(f"  {'step':38s} {(t[:, 3] - t[:, 0]).mean():8.0f}")
print(f"  {'entry -> W_hh in registers':38s} {(t[:, 5] - t[:, 4]).mean():8.0f}")
print(f"  {'all T steps':38s} {(t[:, 6] - t[:, 5]).mean():8.0f}   ({(t[:, 6] - t[:, 5]).mean() / T:.0f} per step)")
print(f"  {'first entry -> last exit':38s} {t[:, 6].max() - t[:, 4].min():8.0f}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    net(x)
e1.record()
torch.cuda.synchronize()
print(f"  whole Brain.forward (3 launches + copy): {e0.elapsed_time(e1) * 50:.1f} us")

# ---- BPTT kernel of one DQN step (B = 128: 256 workgroups, one (direction, sample) row each)
from ivos_w_amd.models.agent import Agent  # noqa: E402
from types import SimpleNamespace as NS  # noqa: E402
cfg = NS(phase="train", data=NS(subset="train"), agent=NS(memory_size=1000, gamma=0.95, eps_start=0.7, eps_end=0.25, eps_decay=500,
                                                         update_rate=0.05, lr=5e-6, weight_decay=5e-4))
agent = Agent(dev, cfg)
tr = synth.replay_transitions(n=600, T=25, seed=3)
batch = synth.collate_np(tr, synth.minibatch_indices(0, n=600, B=128, seed=5))
for _ in range(2):
    agent.loss_and_grads(batch)
tb = torch.zeros(256, 8, dtype=torch.int64, device=dev)
L.lib().ivosw_lstm_probe(None, L.dptr(tb))
agent.loss_and_grads(batch)
torch.cuda.synchronize()
L.lib().ivosw_lstm_probe(None, None)
b = tb.cpu().numpy().astype(np.float64)
b = b[b[:, 6] >= 5]                       # rows with at least 5 steps (the probed step exists)
print(f"BPTT, B=128 T=25: {len(b)} probed workgroups; ticks of one step (mean)")
print(f"  {'gate gradients -> LDS':38s} {(b[:, 1] - b[:, 0]).mean():8.0f}")
print(f"  {'barrier':38s} {(b[:, 2] - b[:, 1]).mean():8.0f}")
print(f"  {'dpre . W_hh (128 fma) + quad sum':38s} {(b[:, 3] - b[:, 2]).mean():8.0f}")
print(f"  {'per step over the whole loop':38s} {((b[:, 5] - b[:, 4]) / b[:, 6]).mean():8.0f}")
