"""Host logic (no GPU): recommendation glue, checkpoint helpers, agent bookkeeping vs known answers recorded from
the reference (tests/golden/glue_fixtures.json)."""
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

from ivos_w_amd.utils import misc
from ivos_w_amd.utils import utils_agent as ua


@pytest.fixture(scope="module")
def fx(golden_dir):
    return json.load(open(os.path.join(golden_dir, "glue_fixtures.json")))


def test_select_next_frame(fx):
    v = np.array([0.7, 0.2, 0.9, 0.1, 0.5, 0.3])
    for case in fx["select"]:
        assert int(ua.select_next_frame(v.copy(), metric=case["metric"], prev_frames=case["prev"])) == case["got"], case
    np.random.seed(4)
    assert [int(ua.select_next_frame(v, metric="random")) for _ in range(5)] == fx["select_random"]


def test_gen_subseq(fx):
    for case in fx["gen_subseq"]:
        assert [int(x) for x in ua.gen_subseq(*case["args"])] == case["got"], case


def test_goal_only_reward(fx):
    df = pd.DataFrame(fx["reward_df"])
    r = fx["reward"]
    step, done = ua.goal_only_reward("bear", 3, 5, False, np.array(r["iou_new"]), df=df)
    assert int(step) == r["step"] and abs(float(done) - r["done"]) < 1e-12
    step2, done2 = ua.goal_only_reward("bear", 3, 5, True, np.array(r["iou_new"]), df=None)
    assert int(step2) == r["step_repeat"] and float(done2) == r["done_nodf"]
    with pytest.raises(AssertionError):                      # exactly 30 baselines are required (reference :20)
        ua.goal_only_reward("camel", 3, 2, False, np.array(r["iou_new"]), df=df)


class _FakeAgent:
    def __init__(self):
        self.pushed, self.updates = [], 0

    def memory(self, *a):
        self.pushed.append(a)

    def update_agent(self, sample):
        self.updates += 1
        return 0.5

    def action(self, state):
        self.last_state = state
        return int(np.argmin(state[:, 0]))


class AD(dict):
    __getattr__ = dict.__getitem__


def test_agent_business_pushes_and_trains(fx):
    df = pd.DataFrame(fx["reward_df"])
    agent = _FakeAgent()
    cfg = AD(phase="train")
    iou = np.array(fx["reward"]["iou_new"])
    loader = [dict(i=i) for i in range(40)]
    loss, rs, rd = ua.agent_business(cfg, agent, 3, 3, False, iou * 0.9, iou, 2, "bear", {"bear": 5}, False, df, [1, 1], 4,
                                     {"m": 0}, {"m": 1}, "/tmp/x", loader)
    assert agent.updates == 3 * 3 - 1 and float(loss) == 0.5 and int(rs) == 1
    a = agent.pushed[0]
    assert a[5] is True                                               # done at the last interaction
    assert a[8] == "0.0/2.0/0.0/0.0/0.0" and a[9] == "0.0/2.0/0.0/0.0/1.0"  # annotated / next-annotated counts as strings
    assert a[6] == "/".join(str(v) for v in iou * 0.9)
    agent2 = _FakeAgent()
    out = ua.agent_business(AD(phase="eval"), agent2, 3, 3, False, iou, iou, 2, "bear", {"bear": 5}, False, df, [1], 4, {}, {},
                            "/tmp/x", loader)
    assert agent2.pushed == [] and [float(x) for x in out] == [0.0, 0.0, 0.0]


def test_recommend_frame_oracle_and_linspace():
    agent = _FakeAgent()
    q = np.array([0.9, 0.3, 0.8, 0.2])
    kw = dict(n_frame=4, n_objects=1, all_F=None, all_P=None, new_masks_quality=q, prev_frames=[3],
              annotated_frames_list=[3, 3], mask_quality=np.zeros(4), first_frame=3, max_nb_interactions=8)
    assert ua.recommend_frame(AD(setting="oracle", method="worst"), None, agent, "cpu", **kw) == 1
    assert ua.recommend_frame(AD(setting="oracle", method="ours"), None, agent, "cpu", **kw) == 3
    np.testing.assert_array_equal(agent.last_state, np.stack([q, [0, 0, 0, 2]], 1))
    assert ua.recommend_frame(AD(setting="wild", method="linspace"), None, agent, "cpu", **kw) in (0, 1, 2)
    with pytest.raises(NotImplementedError):
        ua.recommend_frame(AD(setting="nope", method="ours"), None, agent, "cpu", **kw)


def test_checkpoint_helpers_roundtrip(tmp_path):
    from ivos_w_amd.models.agent import Brain
    net = Brain()
    misc.save_agent_checkpoint(net, str(tmp_path))
    sd = torch.load(tmp_path / "agent.pt")
    assert list(sd.keys()) == list(net.state_dict().keys())

    class Holder:
        policy_net = Brain()
    assert misc.load_agent_checkpoint(Holder, str(tmp_path)) == 1
    for a, b in zip(Holder.policy_net.state_dict().values(), net.state_dict().values()):
        assert torch.equal(a, b)
    assert misc.load_agent_checkpoint(Holder, str(tmp_path / "missing")) is None      # never raises
    assert misc.load_agent_checkpoint(None, str(tmp_path)) is None
    torch.save({"module.encoder_fc1.weight": torch.zeros(3)}, tmp_path / "agent.pt")    # wrong shape -> swallowed
    assert misc.load_agent_checkpoint(Holder, str(tmp_path)) == -1
    assert misc.load_network_checkpoint(str(tmp_path / "nope.pt"), encoder=None) is False


def test_checkpoint_helpers_vs_reference_golden(tmp_path, golden_dir):
    """The checkpoint / meter / seed helpers against the REFERENCE's utils/misc.py:11-115 run in the build container on the same
    scenarios (tests/golden/scenarios.misc_scenarios, recorded by make_goldens.py misc): return values (None / 1 / -1, True /
    False), the key rewrites ('module.' stripped by k[7:] whenever 'module' occurs ANYWHERE in the key, 'encoder.' put in front of
    video_match 'base' keys, 'module.' added only when str(device) == 'cuda' resp. device == 'gpu'), the strict flag passed through,
    what the save helpers leave in their files (the un-stripped state_dict), what is printed, and which loader swallows a failing
    load_state_dict (the agent's) and which does not (the network's)."""
    from tests.golden import scenarios as sc
    want = json.load(open(os.path.join(golden_dir, "misc_fixtures.json")))
    got = json.loads(json.dumps(sc.misc_scenarios(misc, str(tmp_path))))
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], (k, got[k], want[k])


def test_assessnet_checkpoint_roundtrip(tmp_path):
    from ivos_w_amd.models.assessment import AssessNet
    a, b = AssessNet(), AssessNet()
    misc.save_network_checkpoint(str(tmp_path), a)
    assert misc.load_network_checkpoint(str(tmp_path / "assess_net.pt"), encoder=b, device="cpu", strict=True) is True
    assert len(b.state_dict()) == 326
    for x, y in zip(a.state_dict().values(), b.state_dict().values()):
        assert torch.equal(x, y)


def test_brain_flat_arena_views_and_seeded_init():
    from ivos_w_amd.models.agent import Brain
    from ivos_w_amd import synth
    torch.manual_seed(0)
    b = Brain()
    assert b.flat.numel() == 180993 and b.flat_grad.numel() == 180993
    off = synth.brain_offsets()
    for k, p in b.named_parameters():
        o, shp = off[k]
        assert p.data_ptr() == b.flat.data_ptr() + 4 * o and tuple(p.shape) == tuple(shp)
        assert p.grad.data_ptr() == b.flat_grad.data_ptr() + 4 * o
    sd = synth.brain_state_dict(2)
    b.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    np.testing.assert_array_equal(b.flat.numpy(), synth.brain_flat(sd))        # load_state_dict writes through the views


def test_average_meter_and_seed():
    m = misc.AverageMeter()
    m.update(2.0, 2)
    m.update(5.0)
    assert m.avg == 3.0 and m.count == 3
    misc.set_random_seed(3)
    a = np.random.random()
    misc.set_random_seed(3)
    assert a == np.random.random()
