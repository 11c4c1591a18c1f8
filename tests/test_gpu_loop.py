"""GPU integration: the rows of SURVEY §8 composed the way an interaction loop runs them —
segmentation epilogue (get_results, stand-in MANet) -> all_P -> recommend_frame (batched AssessNet over all objects ->
mask quality -> worst-frame rule / Brain argmax) and J&F of the predicted labels — against the same pipeline built from
the oracles (reference control flow of utils/utils_agent.py:104-122 with one AssessNet forward per object)."""
import numpy as np
import pytest
import torch

from ivos_w_amd import synth
from ivos_w_amd.utils import misc, utils_agent, utils_manet
from oracle import assess_oracle as ao
from oracle import brain_oracle as bo
from oracle import jf_oracle as jo
from oracle import seg_oracle as so
from tests.test_gpu_seg_epilogue import FakeMANet

pytestmark = pytest.mark.gpu


class AD(dict):
    __getattr__ = dict.__getitem__


def test_interaction_round_matches_oracle_pipeline():
    dev = torch.device("cuda:0")
    n, O, hs, ws, h, w = 6, 2, 60, 107, 240, 427
    C = O + 1
    g = torch.Generator().manual_seed(11)
    all_F = torch.rand(n, 3, h, w, generator=g)                      # frames stay CPU tensors in the reference until recommend_frame
    emb = torch.randn(n, 8, 6, 6, generator=g)
    args = dict(scribble_label=None, prev_label=None, eval_global_map_tmp_dic={}, local_map_dics=({}, {}), n_interaction=1,
                sequence="seq", obj_nums=O, next_frame=2, first_scribble=True, h=h, w=w, total_frame_num=n)
    # --- product path
    from ivos_w_amd.models.agent import Agent
    from ivos_w_amd.models.assessment import AssessNet
    net = AssessNet(precision="fp32")
    sd = synth.assessnet_state_dict(0)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    net = net.to(dev).eval()
    cfg = AD(phase="eval", data=AD(subset="val"), agent=AD(memory_size=100, gamma=0.95, eps_start=0.7, eps_end=0.25, eps_decay=500,
                                                           update_rate=0.05, lr=5e-6, weight_decay=5e-4))
    agent = Agent(dev, cfg)
    P = synth.brain_state_dict(0)
    agent.policy_net.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    store = utils_manet.ProbStore(n, C, h, w, dev)
    fm, all_P = utils_manet.get_results(FakeMANet(C, hs, ws, dev), emb[2:3].to(dev), prev_label_storage={}, embedding_memory=emb.to(dev),
                                        knns=5, store=store, **args)
    quality = np.zeros(n)
    kw = dict(n_frame=n, n_objects=O, all_F=all_F, all_P=all_P, new_masks_quality=np.zeros(n), prev_frames=[2],
              annotated_frames_list=[2], mask_quality=quality, first_frame=2, max_nb_interactions=8)
    nxt_worst = utils_agent.recommend_frame(AD(setting="wild", method="worst"), net, agent, dev, **kw)
    q_worst = quality.copy()
    nxt_ours = utils_agent.recommend_frame(AD(setting="wild", method="ours"), net, agent, dev, **kw)
    # --- oracle pipeline (reference control flow, CPU)
    fm_o, all_P_o = so.get_results(FakeMANet(C, hs, ws, torch.device("cpu")), emb[2:3], prev_label_storage={}, embedding_memory=emb,
                                   knns=5, **args)
    tsd = ao.to_torch_sd(sd)
    pred = np.zeros((n, O))
    for i in range(O):
        pred[:, i] = ao.assess_forward(tsd, all_F.numpy(), all_P_o[:, i + 1].numpy())
    want_q = pred.mean(1)
    np.testing.assert_allclose(q_worst, want_q, rtol=1e-4)
    np.testing.assert_allclose(quality, want_q, rtol=1e-4)                       # the caller's array is filled in place
    order = want_q.argsort()
    assert nxt_worst == next(i for i in order if i not in [2])
    counts = np.zeros(n)
    counts[2] += 1
    state = np.stack([want_q, counts], 1)
    q = bo.brain_forward(P, state[None].astype(np.float32))
    assert nxt_ours == int(q[0].argmax())                                       # cfg.phase != 'train' -> greedy
    # --- J&F of the predicted labels against a synthetic ground truth, labels never leaving the device
    np.testing.assert_array_equal(fm.cpu().numpy(), fm_o.numpy())
    gt = np.roll(fm_o.numpy().astype(np.uint8), 3, axis=2)
    got = misc.sequence_metric("J_AND_F", torch.from_numpy(gt).to(dev), store.labels_u8, O)
    want = jo.sequence_metric("J_AND_F", gt.astype(np.int64), fm_o.numpy().astype(np.int64), O)
    np.testing.assert_array_equal(got, want)


def _nets(dev, precision="fp32"):
    from ivos_w_amd.models.agent import Agent
    from ivos_w_amd.models.assessment import AssessNet
    net = AssessNet(precision=precision)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.assessnet_state_dict(0).items()}, strict=True)
    net = net.to(dev).eval()
    cfg = AD(phase="eval", data=AD(subset="val"), agent=AD(memory_size=100, gamma=0.95, eps_start=0.7, eps_end=0.25, eps_decay=500,
                                                           update_rate=0.05, lr=5e-6, weight_decay=5e-4))
    agent = Agent(dev, cfg)
    agent.policy_net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.brain_state_dict(0).items()})
    return net, agent


@pytest.mark.parametrize("n,O,layout", [(6, 3, "reference"), (5, 2, "object_major"), (3, 9, "reference")])
def test_device_resident_recommendation(n, O, layout):
    """SURVEY 8(f) row 1 (utils/utils_agent.py:104-122): across 8 interactions the video is uploaded ONCE, every object's
    masks are read in place from all_P (no repeat of the frames), and quality -> state -> Brain -> argmax stays on the
    device.  Checked against the reference control flow run on the product kernels: one AssessNet forward per object,
    float64 `pred.mean(1)` in numpy (O = 9 takes numpy's 8-way pairwise path), `agent.action(state)` from the host."""
    dev = torch.device("cuda:0")
    h, w = 120, 214
    net, agent = _nets(dev)
    g = torch.Generator().manual_seed(100 * n + O)
    all_F = torch.rand(n, 3, h, w, generator=g)                           # CPU, as the entry scripts hold it
    logits = torch.randn(n, O + 1, h // 4, w // 4, generator=g) * 3
    probs = torch.softmax(torch.nn.functional.interpolate(logits, (h, w), mode="bilinear", align_corners=True), 1).to(dev)
    if layout == "object_major":                                          # the ProbStore layout: [C, n, H, W] storage, permuted view
        all_P = probs.permute(1, 0, 2, 3).contiguous().permute(1, 0, 2, 3)
        assert not all_P.is_contiguous()
    else:
        all_P = probs.contiguous()
    utils_agent.clear_frame_cache()
    up0 = utils_agent.frame_cache.uploads
    quality = np.zeros(n)
    annotated, picks, qualities = [1], [], []
    for it in range(8):
        kw = dict(n_frame=n, n_objects=O, all_F=all_F, all_P=all_P, new_masks_quality=np.zeros(n), prev_frames=list(annotated),
                  annotated_frames_list=list(annotated), mask_quality=quality, first_frame=1, max_nb_interactions=8)
        nxt = utils_agent.recommend_frame(AD(setting="wild", method="ours" if it % 2 == 0 else "worst"), net, agent, dev, **kw)
        picks.append((it, int(nxt), list(annotated)))
        qualities.append(quality.copy())
        annotated.append(int(nxt))
    assert utils_agent.frame_cache.uploads - up0 == 1                     # one H2D of the video for the whole sequence
    # reference control flow on the product kernels
    fdev = all_F.to(dev)
    pred = np.zeros((n, O))
    for i in range(O):
        pred[:, i] = net(fdev, all_P[:, i + 1].contiguous()).cpu().numpy().reshape(-1)
    want_q = pred.mean(1)
    for q in qualities:
        np.testing.assert_array_equal(q, want_q)                           # float64 mean, numpy's summation order, bit for bit
    steps0 = agent.steps_done
    for it, nxt, ann in picks:
        counts = np.zeros(n)
        for i in ann:
            counts[i] += 1
        if it % 2 == 0:
            assert nxt == int(agent.action(np.stack([want_q, counts], 1), verbose=False))
        else:
            assert nxt == int(utils_agent.select_next_frame(want_q, metric="worst", prev_frames=ann))
    assert steps_done_ok(agent, steps0)


def steps_done_ok(agent, before):
    return agent.steps_done == before + 4          # four 'ours' interactions re-checked from the host
