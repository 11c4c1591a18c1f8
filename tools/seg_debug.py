import os, sys, torch, numpy as np
sys.path.insert(0, '.')
from ivos_w_amd.utils import utils_manet
from oracle import seg_oracle as so
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(304)
base = torch.randn(3, 4, 20, 35, generator=g) * 3.0
x = torch.nn.functional.interpolate(base, size=(120, 214), mode="bicubic", align_corners=False)
x = (x + 0.3 * torch.randn(3, 4, 120, 214, generator=g)).contiguous()
lab, probs = utils_manet.seg_epilogue(x.to(dev), 480, 854)
up, wl = so.epilogue(x, 480, 854)
wp = torch.softmax(up, 1)
d = (probs.cpu() - wp).abs()
print(os.environ.get("IVOSW_SEG_SCALAR"), "max", d.max().item(), "frac>2e-6", (d > 2e-6).float().mean().item())
i = d.argmax().item(); idx = np.unravel_index(i, d.shape); print(idx, probs.cpu()[idx].item(), wp[idx].item())
# upsampled logits on GPU via torch for comparison
upg = torch.nn.functional.interpolate(x.to(dev), size=(480, 854), mode='bilinear', align_corners=True).cpu()
print("torch gpu vs cpu upsample max diff", (upg - up).abs().max().item())
pg = torch.softmax(upg.to(dev), 1).cpu()
print("torch gpu softmax vs cpu", (pg - wp).abs().max().item(), ((pg - wp).abs() > 2e-6).float().mean().item())
bad = (d > 2e-6).any(1)       # [k, H, W]
ks, ys, xs = np.nonzero(bad.numpy())
if len(ys) == 0:
    print("no pixel deviates by more than 2e-6")
    raise SystemExit(0)
idx = ys * 854 + xs
print("violating pixels:", len(ys), "idx%4 histogram", np.bincount(idx % 4, minlength=4), "x range", xs.min(), xs.max(), "y range", ys.min(), ys.max())
print("first 12 (y,x):", list(zip(ys[:12].tolist(), xs[:12].tolist())))
print("distinct x:", np.unique(xs)[:40], "count", len(np.unique(xs)))
print("distinct y:", np.unique(ys)[:40], "count", len(np.unique(ys)))
