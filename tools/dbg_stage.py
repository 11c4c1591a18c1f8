import sys, os
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from ivos_w_amd import synth, _lib as L
from ivos_w_amd.models.assessment import AssessNet
dev = torch.device("cuda:0")
lib = L.lib()
net = AssessNet(precision="bf16")
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.assessnet_state_dict(0).items()}, strict=True)
net.to(dev).eval()
for B in (8, 3, 16):
    tf, tp = synth.assess_inputs(B, seed=1234 + B, edge_cases=(B == 8), structured=True)
    ttf, ttp = torch.from_numpy(tf).to(dev), torch.from_numpy(tp).to(dev)
    got = {}
    for mode in (1, 0):
        lib.ivosw_tune_set(b"RES2_STAGE", mode)
        got[mode] = [net.forward_tap(ttf, ttp, nm)[1].clone().float() for nm in ("res2", "res3")]
    lib.ivosw_tune_set(b"RES2_STAGE", 0)
    for a, b, nm in zip(got[1], got[0], ("res2", "res3")):
        d = (a - b).abs()
        bad = (d > 0).nonzero()
        print(B, nm, "max", d.max().item(), "nbad", bad.shape[0], "of", d.numel())
        if bad.shape[0]:
            fb = bad[:, 0].unique().tolist(); ys = bad[:, 1].unique().tolist(); xs = bad[:, 2].unique().tolist(); cs = bad[:, 3].unique().tolist()
            print("  frames", fb[:10], "rows", ys[:40], "cols", xs[:40], "nch", len(cs), cs[:16])
# per-tile bad counts for the last case (res2)
a, b2 = got[1][0], got[0][0]
d = ((a - b2).abs() > 0).float()          # [B,64,64,256]
t = d.view(-1, 8, 8, 4, 16, 256).sum(dim=(2, 4, 5))   # [B, tile row, tile col]
print("bad elements per 8x16 tile (of 32768), frame 0:\n", t[0].int().cpu().numpy())
print("frame 1:\n", t[1].int().cpu().numpy())
ch = d[0].sum(dim=(0, 1))
print("bad per channel (frame 0), first 64:", ch[:64].int().tolist())
pix = d[0, :8, :16].sum(dim=2)
print("tile (0,0) bad per pixel:\n", pix.int().cpu().numpy())
