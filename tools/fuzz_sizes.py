import numpy as np, torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ivos_w_amd import synth
from ivos_w_amd.models.assessment import AssessNet
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.assessnet_state_dict(0).items()}
nets = {}
for prec in ("bf16", "bf16x3", "fp32"):
    n = AssessNet(precision=prec); n.load_state_dict(sd); n.to(dev).eval(); nets[prec] = n
ok = True
for B in (1, 2, 3, 7, 33, 65, 100, 129, 257):
    tf, tp = synth.assess_inputs(B, seed=100 + B)
    ttf, ttp = torch.from_numpy(tf).to(dev), torch.from_numpy(tp).to(dev)
    s = {p: nets[p](ttf, ttp).cpu().numpy().reshape(-1) for p in nets}
    e_x3 = np.abs(s["bf16x3"] / s["fp32"] - 1).max(); e_bf = np.abs(s["bf16"] / s["fp32"] - 1).max()
    # batch independence: frame 0 alone
    s1 = nets["bf16x3"](ttf[:1], ttp[:1]).cpu().numpy().reshape(-1)[0]
    s1b = nets["bf16"](ttf[:1], ttp[:1]).cpu().numpy().reshape(-1)[0]
    print(f"B={B:4d}: x3 vs fp32 {e_x3:.2e}  bf16 vs fp32 {e_bf:.2e}  frame0 alone == in batch: x3 {s1 == s['bf16x3'][0]} bf16 {s1b == s['bf16'][0]}")
    ok &= e_x3 < 2e-5 and e_bf < 4e-3 and np.isfinite(s["bf16"]).all()
print("FUZZ", "OK" if ok else "FAILED")
