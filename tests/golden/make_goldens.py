"""Golden-vector generator.  Runs ONLY in the build container: it imports the reference
(/root/reference, read-only) with the shims of SURVEY.md Appendix D, feeds it the seeded synthetic
weights/inputs of ``ivos_w_amd.synth`` and records the reference's own outputs as small .npz/.json
fixtures next to this file.  The fixtures are data; no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py [brain] [dqn] [assess] [replay] [glue] [action] [seg] [misc]
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
    # S8: the B8 inputs (edge masks included) through the reference with the SPREAD weight recipe (synth.assessnet_state_dict(0,
    # spread=True)): the default recipe's scores lie within 4 % of each other, so a 1e-4 check on them has little power against,
    # say, a wrong tap in res5; the spread recipe's scores differ by far more than any tolerance from frame to frame
    net_spread = AssessNet()
    net_spread.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.assessnet_state_dict(0, spread=True).items()}, strict=True)
    net_spread.eval()
    net_default = net
    for tag, B, edge in (("B8", 8, True), ("B1", 1, False), ("B3", 3, False), ("S8", 8, True)):
        net = net_spread if tag == "S8" else net_default
        tf, tp = synth.assess_inputs(B, seed=1234 + (8 if tag == "S8" else B), edge_cases=edge, structured=True)
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



# ----------------------------------------------------------------------------- replay memory / sampler
def fixture_rows(n=9, T=4, seed=3):
    """Deterministic transitions in the reference's CSV cell format ('/'-joined str(float))."""
    rs = np.random.RandomState(seed)
    seqs = ["bear", "camel", "drift", "bear", "camel", "elephant", "drift", "bear", "flamingo"]
    rows = []
    for i in range(n):
        iou = np.round(rs.uniform(0.5, 0.8, T), 4)
        if seqs[i] == "camel":                                # camel never improves by > 0.05: filtered by sample_th
            iou = np.full(T, 0.3)
        gain = 0.01 if seqs[i] == "camel" else 0.1
        nxt = np.round(np.clip(iou + gain, 0, 1), 4)
        ann = np.zeros(T)
        ann[rs.randint(T)] += 1
        act = int(rs.randint(T))
        nann = ann.copy()
        nann[act] += 1
        j = lambda a: "/".join(str(float(v)) for v in a)
        rows.append(dict(sequence=seqs[i], scribble_iter=1 + i % 3, n_interaction=1 + i % 4, n_interaction_next=2 + i % 4,
                         action=act, reward_step=1 if i % 3 else -1, reward_done=float(np.round(rs.randn(), 5)),
                         done=bool(i % 4 == 3), state_iou=j(iou), next_state_iou=j(nxt), annotated_frames=j(ann),
                         next_annotated_frames=j(nann)))
    return rows


def make_replay():
    import shutil
    import tempfile
    from models.momory_pool import ReplayMemory
    from datasets.agent_dataset import DAVIS2017AgentTrain
    out = {}
    rows = fixture_rows()
    tmp = tempfile.mkdtemp(prefix="ivosw_gold_")
    try:
        # (1) push / push_to_csv with capacity 5 (ring wraps, oldest CSV rows dropped)
        mem = ReplayMemory(5)
        d1 = os.path.join(tmp, "a")
        os.makedirs(d1)
        for r in rows[:7]:
            st = dict(sequence=r["sequence"], scribble_iter=r["scribble_iter"], n_interaction=r["n_interaction"])
            nst = dict(sequence=r["sequence"], scribble_iter=r["scribble_iter"], n_interaction=r["n_interaction_next"])
            mem.push(st, r["action"], nst, r["reward_step"], r["reward_done"], r["done"], r["state_iou"],
                     r["next_state_iou"], r["annotated_frames"], r["next_annotated_frames"])
            mem.push_to_csv(d1)
        out["push_csv"] = open(os.path.join(d1, "memory_pool.csv")).read()
        out["push_position"] = mem.position
        out["push_len"] = len(mem)
        out["push_actions_in_ring"] = [int(t.action) for t in mem.memory]
        # (2) load_from_csv with the per-sequence filter
        import pandas as pd
        src = os.path.join(tmp, "pretrain.csv")
        pd.DataFrame(rows, columns=mem.COLUMNS).to_csv(src)
        out["pretrain_csv"] = open(src).read()
        mem2 = ReplayMemory(8)                       # capacity truncates the 9 rows to 8 first
        d2 = os.path.join(tmp, "b")
        mem2.load_from_csv(src, d2, sample_th=0.05)
        out["load_seq_list"] = list(mem2.seq_list)
        out["load_capacity"] = int(mem2.capacity)
        out["load_len"] = len(mem2)
        out["load_position"] = int(mem2.position)
        out["load_csv"] = open(os.path.join(d2, "memory_pool.csv")).read()
        out["load_actions"] = [int(t.action) for t in mem2.memory]
        # (3) the real minibatch source: dataset + DataLoader collation
        root = os.path.join(tmp, "DAVIS")
        os.makedirs(os.path.join(root, "ImageSets", "2017"))
        with open(os.path.join(root, "ImageSets", "2017", "train.txt"), "w") as f:
            f.write("\n".join(sorted({r["sequence"] for r in rows})) + "\n")
        np.random.seed(0)
        ds = DAVIS2017AgentTrain(split="train", db_root_dir=root, save_result_dir=d2, memory_size=100,
                                 seq_list=mem2.seq_list)
        loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False)
        batch = next(iter(loader))
        out["ds_len"] = len(ds)
        out["batch"] = {k: dict(dtype=str(v.dtype), shape=list(v.shape), values=v.double().numpy().tolist())
                        for k, v in batch.items()}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(HERE, "replay_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("replay_fixtures.json", {k: (v if not isinstance(v, (str, dict)) else "...") for k, v in out.items()})


# ----------------------------------------------------------------------------- recommendation glue known answers
def make_glue():
    import pandas as pd
    from utils import utils_agent as ua
    out = {}
    v = np.array([0.7, 0.2, 0.9, 0.1, 0.5, 0.3])
    out["select"] = [
        dict(metric="worst", prev=[3], got=int(ua.select_next_frame(v.copy(), metric="worst", prev_frames=[3]))),
        dict(metric="worst", prev=[3, 1, 5], got=int(ua.select_next_frame(v.copy(), metric="worst", prev_frames=[3, 1, 5]))),
        dict(metric="worst", prev=list(range(6)), got=int(ua.select_next_frame(v.copy(), metric="worst", prev_frames=list(range(6))))),
        dict(metric="max", prev=[2], got=int(ua.select_next_frame(v.copy(), metric="max", prev_frames=[2]))),
        dict(metric="min", prev=None, got=int(ua.select_next_frame(v.copy(), metric="min"))),
    ]
    np.random.seed(4)
    out["select_random"] = [int(ua.select_next_frame(v, metric="random")) for _ in range(5)]
    out["gen_subseq"] = [dict(args=list(a), got=[int(x) for x in ua.gen_subseq(*a)]) for a in [
        (10, 50, 25, "consecutive"), (0, 50, 25, "consecutive"), (49, 50, 25, "consecutive"), (30, 40, 25, "consecutive"),
        (7, 50, 8, "equal"), (0, 50, 8, "equal"), (49, 50, 8, "equal"), (3, 6, 8, "equal"), (20, 104, 8, "equal")]]
    # goal_only_reward: 30 random-policy baselines for (sequence, n_interaction_next, scribble slot)
    rs = np.random.RandomState(8)
    rows = []
    for k in range(30):
        rows.append(dict(sequence="bear", n_interaction_next=3, scribble_iter=2 + 3 * k,
                         next_state_iou="/".join(str(float(x)) for x in np.round(rs.uniform(0.4, 0.8, 5), 4))))
    for k in range(10):                                   # distractors: other slot / interaction / sequence
        rows.append(dict(sequence="bear", n_interaction_next=3, scribble_iter=1 + 3 * k, next_state_iou="0.1/0.1"))
        rows.append(dict(sequence="bear", n_interaction_next=4, scribble_iter=2, next_state_iou="0.2/0.2"))
        rows.append(dict(sequence="camel", n_interaction_next=3, scribble_iter=2, next_state_iou="0.3/0.3"))
    df = pd.DataFrame(rows)
    out["reward_df"] = rows
    iou_new = np.round(rs.uniform(0.5, 0.9, 5), 4)
    r_step, r_done = ua.goal_only_reward("bear", 3, 5, False, iou_new, df=df)
    r_step2, r_done2 = ua.goal_only_reward("bear", 3, 5, True, iou_new, df=None)
    out["reward"] = dict(iou_new=iou_new.tolist(), step=int(r_step), done=float(r_done), step_repeat=int(r_step2),
                         done_nodf=float(r_done2))
    with open(os.path.join(HERE, "glue_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("glue_fixtures.json", out["select"], out["reward"]["done"])

# ----------------------------------------------------------------------------- Agent.action, train phase
def make_action():
    """40 consecutive Agent.action calls of the reference in the TRAIN phase (epsilon-greedy, models/agent.py:168-196): seeded
    python / numpy RNGs, the seeded Brain weights, a fresh state per call.  Records every returned index, which branch took it
    and the threshold — the greedy picks pin the bit-exact argmax, the random ones the RNG call sequence."""
    import io
    import random
    import contextlib
    from models.agent import Agent
    agent = Agent(torch.device("cpu"), agent_cfg(phase="train"))
    agent.policy_net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.brain_state_dict(0).items()})
    random.seed(1234)
    np.random.seed(1234)
    out = dict(T=25, seed=1234, calls=[])
    for i in range(40):
        state = brain_inputs(1, 25, 500 + i)[0].astype(np.float64)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            a = agent.action(state)
        out["calls"].append(dict(action=int(a), random="randomly" in buf.getvalue(), steps_done=int(agent.steps_done)))
    with open(os.path.join(HERE, "agent_action.json"), "w") as f:
        json.dump(out, f)
    print("agent_action.json", sum(c["random"] for c in out["calls"]), "random of", len(out["calls"]))


# ----------------------------------------------------------------------------- seg: utils/utils_manet.py:59-163 (get_results)
def make_seg():
    """Runs the REFERENCE's get_results (imported from /root/reference/utils/utils_manet.py) on the CPU with the deterministic
    stand-in model of tests/golden/scenarios.py.  Shims beyond Appendix D: `davisinteractive.dataset.davis` (the module imports
    `Davis` from it, :6; unused by get_results), `config` (a module whose `cfg.KNNS` the propagation calls read, :101) and
    `torch.Tensor.cuda` = identity for the duration of the calls (`prev_label = prev_label.cuda()`, :90, :123)."""
    from tests.golden import scenarios as sc
    ddd = types.ModuleType("davisinteractive.dataset.davis")
    ddd.Davis = sys.modules["davisinteractive.dataset"].Davis
    sys.modules["davisinteractive.dataset.davis"] = ddd
    cfgmod = types.ModuleType("config")
    cfgmod.cfg = AD(KNNS=sc.SEG_KNNS)
    had = sys.modules.get("config")
    sys.modules["config"] = cfgmod
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        from utils import utils_manet as ref
        assert ref.__file__.startswith(REF), ref.__file__
        out = {}
        for name in sc.SEG_CASES:
            model, kw = sc.seg_case(name, torch.device("cpu"))
            store = {}
            with torch.no_grad():
                fm, ap = ref.get_results(model, kw["ref_frame_embedding"], kw["scribble_label"], kw["prev_label"], kw["eval_global_map_tmp_dic"],
                                         kw["local_map_dics"], kw["n_interaction"], kw["sequence"], kw["obj_nums"], kw["next_frame"],
                                         kw["first_scribble"], kw["h"], kw["w"], store, kw["total_frame_num"], kw["embedding_memory"])
            # the stand-in's logits are smooth: the two best upsampled classes never come within 1e-4 of each other, so the label
            # maps of an implementation that agrees to ~1e-6 must be IDENTICAL (recorded so the tests may rely on it)
            top2 = torch.topk(torch.log(ap), 2, dim=1).values
            out[f"{name}.min_top2_gap"] = np.array(float((top2[:, 0] - top2[:, 1]).min()))
            out.update(sc.seg_record(name, fm, ap, store, model.calls))
            print(name, tuple(fm.shape), tuple(ap.shape), "min top-2 gap", float(out[f"{name}.min_top2_gap"]), "calls", model.calls[:3])
    finally:
        torch.Tensor.cuda = real_cuda
        if had is None:
            sys.modules.pop("config", None)
        else:
            sys.modules["config"] = had
    np.savez_compressed(os.path.join(HERE, "seg_get_results.npz"), **out)
    print("seg_get_results.npz", os.path.getsize(os.path.join(HERE, "seg_get_results.npz")), "bytes")


# ----------------------------------------------------------------------------- misc: utils/misc.py:11-115
def make_misc():
    """Runs the checkpoint / meter / seed scenarios of tests/golden/scenarios.py against the REFERENCE's utils/misc.py.  Shims: the
    empty `cv2` of Appendix D and `davisinteractive.metrics.jaccard` with the two names the module imports (:8; only
    sequence_metric, which is not recorded here, calls them)."""
    import tempfile
    from tests.golden import scenarios as sc
    dm, dmj = types.ModuleType("davisinteractive.metrics"), types.ModuleType("davisinteractive.metrics.jaccard")
    dmj.batched_f_measure = dmj.batched_jaccard = None
    dm.jaccard = dmj
    sys.modules["davisinteractive.metrics"], sys.modules["davisinteractive.metrics.jaccard"] = dm, dmj
    from utils import misc as ref
    assert ref.__file__.startswith(REF), ref.__file__
    with tempfile.TemporaryDirectory() as tmp:
        rec = sc.misc_scenarios(ref, tmp)
    json.dump(rec, open(os.path.join(HERE, "misc_fixtures.json"), "w"), indent=1)
    print("misc_fixtures.json", len(rec), "scenarios")


if __name__ == "__main__":
    which = sys.argv[1:] or ["brain", "dqn", "assess", "replay", "glue", "action", "seg", "misc"]
    os.chdir("/tmp")
    install_shims()
    torch.set_num_threads(8)
    for w in which:
        fn = globals().get("make_" + w)
        if fn is None:
            print("skip", w)
            continue
        fn()
