"""Golden-vector generator.  Runs ONLY in the build container: it imports the reference
(/root/reference, read-only) with the shims of SURVEY.md Appendix D, feeds it the seeded synthetic
weights/inputs of ``ivos_w_amd.synth`` and records the reference's own outputs as small .npz/.json
fixtures next to this file.  The fixtures are data; no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py [brain] [dqn] [assess] [replay] [glue]
"""
import json
import os
import sys
import types
from collections import OrderedDict

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

import numpy as np

np.float = float  # models/momory_pool.py:64,66 uses the removed alias
import torch
import torch.nn as nn

sys.path.insert(0, ROOT)
from ivos_w_amd import synth  # noqa: E402


# ----------------------------------------------------------------------------- shims
class _Bottleneck(nn.Module):
    """torchvision ResNet-50 v1.5 bottleneck (stride on the 3x3), own restatement for the stub."""

    def __init__(self, inplanes, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if down:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idt)


class _ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        for li, (_n, nblk, planes, stride) in enumerate(synth.RESNET50_BLOCKS, 1):
            blocks = []
            for b in range(nblk):
                blocks.append(_Bottleneck(inpl, planes, stride if b == 0 else 1, b == 0))
                inpl = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*blocks))


def install_shims():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
    tvm.resnet50 = lambda pretrained=False: _ResNet50()
    tv.models = tvm
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tvm
    di, dd = types.ModuleType("davisinteractive"), types.ModuleType("davisinteractive.dataset")

    class _Davis:
        class _D(dict):
            def __missing__(self, k):
                return {"num_objects": 1}
        dataset = _D()
    dd.Davis = _Davis
    di.dataset = dd
    sys.modules["davisinteractive"], sys.modules["davisinteractive.dataset"] = di, dd
    sys.path.insert(0, REF)


class AD(dict):
    __getattr__ = dict.__getitem__


def agent_cfg(update_rate=0.05, phase="eval"):
    return AD(phase=phase, data=AD(subset="train"),
              agent=AD(memory_size=100000, gamma=0.95, eps_start=0.7, eps_end=0.25, eps_decay=500,
                       update_rate=update_rate, lr=5e-6, weight_decay=5e-4))


def stats(t):
    a = np.asarray(t, np.float64)
    return np.array([a.sum(), np.abs(a).sum()])


# ----------------------------------------------------------------------------- brain
brain_inputs = synth.brain_inputs


def make_brain():
    from models.agent import Brain
    net = Brain()
    sd = synth.brain_state_dict(0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    out = {}
    for i, (N, T) in enumerate([(1, 25), (1, 37), (1, 104), (128, 25), (3, 1), (2, 2)]):
        x = brain_inputs(N, T, 100 + i)
        with torch.no_grad():
            q = net(torch.Tensor(x)).numpy()
        out[f"q_{N}_{T}"] = q
        out[f"argmax_{N}_{T}"] = q.argmax(1)
    np.savez(os.path.join(HERE, "brain_forward.npz"), **out)
    print("brain_forward.npz", {k: v.shape for k, v in out.items()})


# ----------------------------------------------------------------------------- dqn
def collate(tr, idx):
    """What DataLoader default_collate yields for datasets/agent_dataset.py items."""
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.collate_np(tr, idx).items()}


def make_dqn():
    from models.agent import Agent
    tr = synth.replay_transitions(n=2000, T=25, seed=2019)
    out = {}
    for B in (32, 128):
        torch.manual_seed(0)
        agent = Agent(torch.device("cpu"), agent_cfg(update_rate=0.5, phase="train"))
        sd = synth.brain_state_dict(0)
        agent.policy_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        sd_t = synth.brain_state_dict(1)       # a *different* target so Double-DQN is exercised
        agent.target_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_t.items()})
        np.random.seed(5)
        st = np.random.get_state()
        coins = [np.random.random() for _ in range(3)]
        np.random.set_state(st)
        out[f"coins_B{B}"] = np.array(coins)
        for step in range(3):
            idx = synth.minibatch_indices(step, n=2000, B=B, seed=7)
            before = {k: v.detach().double().clone() for k, v in agent.policy_net.state_dict().items()}
            loss = agent.update_agent(collate(tr, idx))
            out[f"loss_B{B}_s{step}"] = np.float64(loss)
            for k, p in agent.policy_net.named_parameters():
                tag = f"B{B}_s{step}_{k}"
                g = p.grad.detach().numpy()
                out["gstat_" + tag] = stats(g)
                out["gslice_" + tag] = g.ravel()[:64].copy()
                d = (p.detach().double() - before[k]).numpy()
                out["dstat_" + tag] = stats(d)
                out["dslice_" + tag] = d.ravel()[:64].copy()
                stt = agent.optimizer.state[p]
                out["m_" + tag] = stt["exp_avg"].numpy().ravel()[:32].copy()
                out["v_" + tag] = stt["exp_avg_sq"].numpy().ravel()[:32].copy()
            synced = all(torch.equal(a, b) for a, b in zip(agent.policy_net.state_dict().values(),
                                                           agent.target_net.state_dict().values()))
            out[f"synced_B{B}_s{step}"] = np.bool_(synced)
        out[f"final_B{B}"] = np.concatenate([v.numpy().ravel() for v in agent.policy_net.state_dict().values()])[::97].copy()
    np.savez(os.path.join(HERE, "dqn_steps.npz"), **out)
    print("dqn_steps.npz", len(out), "arrays;",
          {k: float(v) for k, v in out.items() if k.startswith("loss")},
          {k: bool(v) for k, v in out.items() if k.startswith("synced")})


# ----------------------------------------------------------------------------- assess
def tap_record(out, name, t):
    a = t.detach().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    out["stat_" + name] = np.stack([stats(a[b]) for b in range(a.shape[0])])
    if a.ndim == 4:
        hs, ws = max(1, a.shape[2] // 4), max(1, a.shape[3] // 4)
        out["slice_" + name] = a[:, :6, ::hs, ::ws].copy()
    else:
        out["slice_" + name] = a.reshape(a.shape[0], -1)[:, :64].copy()


def make_assess():
    from models.assessment import AssessNet
    net = AssessNet()
    sd = synth.assessnet_state_dict(0)
    ref_keys = list(net.state_dict().keys())
    assert ref_keys == list(sd.keys()), "state_dict key order mismatch"
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    net.eval()
    with open(os.path.join(HERE, "assessnet_keys.json"), "w") as f:
        json.dump([[k, list(v.shape)] for k, v in net.state_dict().items()], f)
    out = {}
    for tag, B, edge in (("B8", 8, True), ("B1", 1, False), ("B3", 3, False)):
        tf, tp = synth.assess_inputs(B, seed=1234 + B, edge_cases=edge, structured=True)
        taps = {}
        hooks = []
        enc = net.Encoder
        hooks.append(enc.relu.register_forward_hook(lambda m, i, o: taps.setdefault("stem", o.clone())))
        hooks.append(enc.maxpool.register_forward_hook(lambda m, i, o: taps.__setitem__("pool", o.clone())))
        for nm in ("res2", "res3", "res4", "res5"):
            hooks.append(getattr(enc, nm).register_forward_hook(lambda m, i, o, nm=nm: taps.__setitem__(nm, o.clone())))
        hooks.append(enc.register_forward_hook(lambda m, i, o: taps.__setitem__("rois", (i[0].clone(), i[1].clone()))))
        with torch.no_grad():
            ttf, ttp = torch.from_numpy(tf), torch.from_numpy(tp)
            tm = (ttp > 0.5).float()
            yxhw = net.all2yxhw(tm, scale=1.5)
            _fw, _bw, theta = net.get_ROI_grid(yxhw, src_size=tf.shape[2:], dst_size=(256, 256), scale=1.0)
            score = net(ttf, ttp)
        for h in hooks:
            h.remove()
        out[f"{tag}_yxhw"] = yxhw.numpy()
        out[f"{tag}_theta"] = theta.numpy()
        out[f"{tag}_score"] = score.numpy()
        if tag != "B3":
            tap_record(out, f"{tag}_froi", taps["rois"][0])
            tap_record(out, f"{tag}_proi", taps["rois"][1][:, None])
            for nm in ("stem", "pool", "res2", "res3", "res4", "res5"):
                tap_record(out, f"{tag}_{nm}", taps[nm])
        print(tag, "score", score.numpy().ravel())
    np.savez_compressed(os.path.join(HERE, "assess_forward.npz"), **out)
    print("assess_forward.npz", sum(v.nbytes for v in out.values()), "bytes")


if __name__ == "__main__":
    which = sys.argv[1:] or ["brain", "dqn", "assess", "replay", "glue"]
    os.chdir("/tmp")
    install_shims()
    torch.set_num_threads(8)
    for w in which:
        fn = globals().get("make_" + w)
        if fn is None:
            print("skip", w)
            continue
        fn()
