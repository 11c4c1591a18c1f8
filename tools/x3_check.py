"""bf16x3 mode (IVOSW_F32X3) against the fp32 mode and the oracle on a few frames, and frames/s of both at B = 256 (GPU only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivos_w_amd import synth
from ivos_w_amd.models.assessment import AssessNet
from oracle import assess_oracle as ao
dev = torch.device("cuda:0")
sd_np = synth.assessnet_state_dict(0)
nets = {}
for prec in ("fp32", "bf16x3", "bf16"):
    n = AssessNet(precision=prec)
    n.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})
    nets[prec] = n.to(dev).eval()
tf, tp = synth.assess_inputs(8, seed=1234, structured=True)
ref = ao.assess_forward(ao.to_torch_sd(sd_np), tf, tp)
ttf, ttp = torch.from_numpy(tf).to(dev), torch.from_numpy(tp).to(dev)
for prec, n in nets.items():
    got = n(ttf, ttp).cpu().numpy().reshape(-1)
    print(prec, "max rel err vs oracle", float(np.abs(got - ref).max() / np.abs(ref).max()), float((np.abs(got - ref) / np.abs(ref)).max()))
B = 256
tfb, tpb = ttf.repeat(B // 8, 1, 1, 1).contiguous(), ttp.repeat(B // 8, 1, 1).contiguous()
for prec in ("fp32", "bf16x3"):
    n = nets[prec]
    for _ in range(3): n(tfb, tpb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): n(tfb, tpb)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(prec, f"{dt*1e3:.2f} ms per 256 frames = {B/dt:.0f} frames/s")
