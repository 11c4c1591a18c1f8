"""Pins oracle/brain_oracle.py against goldens recorded from the imported reference
(tests/golden/make_goldens.py) and cross-checks the hand-derived BPTT against torch autograd."""
import os

import numpy as np
import pytest
import torch

from ivos_w_amd import synth
from oracle import brain_oracle as bo

CFG = dict(gamma=0.95, lr=5e-6, weight_decay=5e-4, update_rate=0.5)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "brain_forward.npz")), np.load(os.path.join(golden_dir, "dqn_steps.npz"))


@pytest.mark.parametrize("i,N,T", [(0, 1, 25), (1, 1, 37), (2, 1, 104), (3, 128, 25), (4, 3, 1), (5, 2, 2)])
def test_forward_matches_reference(gold, i, N, T):
    P = synth.brain_state_dict(0)
    x = synth.brain_inputs(N, T, 100 + i).astype(np.float32)
    q = bo.brain_forward(P, x)
    np.testing.assert_allclose(q, gold[0][f"q_{N}_{T}"], rtol=1e-4, atol=1e-6)
    assert np.array_equal(q.argmax(1), gold[0][f"argmax_{N}_{T}"])       # bit-exact recommended frame


def _torch_forward(P, x):
    """Independent torch-autograd restatement of the same forward (for the gradient cross-check)."""
    N, T, _ = x.shape
    a1 = torch.relu(x @ P["encoder_fc1.weight"].T + P["encoder_fc1.bias"])
    e = a1 @ P["encoder_fc2.weight"].T + P["encoder_fc2.bias"]
    outs = []
    for order in (range(T), range(T - 1, -1, -1)):
        h = torch.zeros(N, 128, dtype=x.dtype)
        c = torch.zeros(N, 128, dtype=x.dtype)
        hs = [None] * T
        for t in order:
            pre = e[:, t] @ P["lstm_cell.weight_ih"].T + h @ P["lstm_cell.weight_hh"].T
            i, f, g, o = pre.chunk(4, 1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            hs[t] = h
        outs.append(torch.stack(hs, 1))
    hc = torch.relu(torch.cat(outs, 2))
    d1 = torch.relu(hc @ P["decoder_fc1.weight"].T + P["decoder_fc1.bias"])
    return (d1 @ P["decoder_fc2.weight"].T + P["decoder_fc2.bias"])[..., 0]


def test_backward_matches_autograd_fp64():
    P = synth.brain_state_dict(0)
    x = synth.brain_inputs(5, 9, 3)
    rs = np.random.RandomState(1)
    dq = rs.standard_normal((5, 9))
    q, cache = bo.brain_forward(P, x, np.float64, keep=True)
    G = bo.brain_backward(P, cache, dq, np.float64)
    Pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items()}
    qt = _torch_forward(Pt, torch.tensor(x))
    np.testing.assert_allclose(q, qt.detach().numpy(), rtol=1e-10, atol=1e-12)
    (qt * torch.tensor(dq)).sum().backward()
    for k in P:
        np.testing.assert_allclose(G[k], Pt[k].grad.numpy(), rtol=1e-8, atol=1e-12, err_msg=k)


@pytest.mark.parametrize("B", [32, 128])
def test_three_update_steps_match_reference(gold, B):
    g = gold[1]
    tr = synth.replay_transitions(n=2000, T=25, seed=2019)
    P = {k: v.copy() for k, v in synth.brain_state_dict(0).items()}
    Pt = {k: v.copy() for k, v in synth.brain_state_dict(1).items()}
    M = {k: np.zeros_like(v) for k, v in P.items()}
    V = {k: np.zeros_like(v) for k, v in P.items()}
    for step in range(3):
        batch = synth.collate_np(tr, synth.minibatch_indices(step, n=2000, B=B, seed=7))
        before = {k: v.astype(np.float64) for k, v in P.items()}
        loss, G = bo.dqn_loss_and_grads(P, Pt, batch, CFG["gamma"])
        np.testing.assert_allclose(loss, g[f"loss_B{B}_s{step}"], rtol=1e-4)
        for k in P:
            tag = f"B{B}_s{step}_{k}"
            gc = np.clip(G[k], -1, 1)
            np.testing.assert_allclose(gc.ravel()[:64], g["gslice_" + tag], rtol=2e-3, atol=2e-6, err_msg=tag)
            s = np.array([gc.astype(np.float64).sum(), np.abs(gc.astype(np.float64)).sum()])
            np.testing.assert_allclose(s[1], g["gstat_" + tag][1], rtol=1e-3, err_msg=tag)
        bo.clamp_adam(P, G, M, V, step + 1, CFG["lr"], CFG["weight_decay"])
        coin = g[f"coins_B{B}"][step]
        if coin < CFG["update_rate"]:
            Pt = {k: v.copy() for k, v in P.items()}
        assert (coin < CFG["update_rate"]) == bool(g[f"synced_B{B}_s{step}"])
        for k in P:
            tag = f"B{B}_s{step}_{k}"
            d = P[k].astype(np.float64) - before[k]
            np.testing.assert_allclose(d.ravel()[:64], g["dslice_" + tag], rtol=5e-3, atol=2e-8, err_msg=tag)
            np.testing.assert_allclose(M[k].ravel()[:32], g["m_" + tag], rtol=2e-3, atol=1e-7, err_msg=tag)
            np.testing.assert_allclose(V[k].ravel()[:32], g["v_" + tag], rtol=4e-3, atol=1e-12, err_msg=tag)
    flat = synth.brain_flat(P)[::97]
    np.testing.assert_allclose(flat, g[f"final_B{B}"], rtol=1e-4, atol=1e-7)


def test_epsilon_schedule():
    assert abs(bo.epsilon(0, 0.7, 0.25, 500) - 0.7) < 1e-12
    assert abs(bo.epsilon(1000, 0.7, 0.25, 500) - (0.25 + 0.45 * np.exp(-1.0))) < 1e-12


@pytest.mark.parametrize("B", [32])
def test_torch_cpu_baseline_matches_reference_golden(gold, B):
    """bench.py's DQN cpu_baseline (stock torch-CPU ops, own module definitions) against the imported reference's goldens:
    forward Q-values / argmax and three update steps (loss, parameter deltas, sync decisions)."""
    from oracle.torch_cpu_baseline import TorchBrain, TorchDQN
    gf, g = gold
    net = TorchBrain()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.brain_state_dict(0).items()})
    x = synth.brain_inputs(1, 37, 101)
    with torch.no_grad():
        q = net(torch.Tensor(x)).numpy()
    np.testing.assert_allclose(q, gf["q_1_37"], rtol=1e-5, atol=1e-7)
    assert np.array_equal(q.argmax(1), gf["argmax_1_37"])
    tr = synth.replay_transitions(n=2000, T=25, seed=2019)
    dqn = TorchDQN(synth.brain_state_dict(0), synth.brain_state_dict(1), CFG["gamma"], CFG["lr"], CFG["weight_decay"], CFG["update_rate"])
    for step in range(3):
        batch = synth.collate_np(tr, synth.minibatch_indices(step, n=2000, B=B, seed=7))
        before = {k: v.detach().double().numpy().copy() for k, v in dqn.policy.state_dict().items()}
        loss = dqn.update(batch, g[f"coins_B{B}"][step])
        np.testing.assert_allclose(loss, g[f"loss_B{B}_s{step}"], rtol=1e-5)
        for k, v in dqn.policy.state_dict().items():
            d = v.detach().double().numpy() - before[k]
            np.testing.assert_allclose(d.ravel()[:64], g[f"dslice_B{B}_s{step}_{k}"], rtol=2e-3, atol=2e-8, err_msg=k)
        synced = all(torch.equal(a, b) for a, b in zip(dqn.policy.state_dict().values(), dqn.target.state_dict().values()))
        assert synced == bool(g[f"synced_B{B}_s{step}"])
