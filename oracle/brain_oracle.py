"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement, in numpy, of the reference's Double-DQN agent arithmetic:

  * ``brain_forward``      <- Brain.forward                 /root/reference/models/agent.py:33-64
  * ``dqn_targets``/``dqn_loss_and_grads`` <- Agent.update_agent  models/agent.py:107-155
      (the backward pass is hand-derived BPTT; the reference gets it from autograd — the two are
       cross-checked in tests/test_oracle_brain.py against a torch-autograd restatement and against
       goldens recorded from the imported reference)
  * ``clamp_adam``         <- grad clamp + optim.Adam.step   models/agent.py:157-160, :101
  * ``dqn_step``           <- the whole update incl. target sync decision  models/agent.py:103-166

Pinned by tests/golden/brain_*.npz and dqn_*.npz (made by tests/golden/make_goldens.py, which imports
the reference itself in the build container).  dtype is a parameter: fp32 mirrors the reference,
fp64 is used by tests as a high-precision arbiter.
"""
import numpy as np

H = 128  # hidden / lstm input / fc width (models/agent.py:14)


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def brain_forward(P, x, dtype=np.float32, keep=False):
    """P: dict name->ndarray (reference state_dict keys); x [N,T,2] -> Q [N,T].

    The shared LSTMCell(bias=False) runs once forward and once backward over the T frames from a
    zero state (models/agent.py:43-54); gate order i,f,g,o.  ``keep`` returns every intermediate the
    hand-derived backward needs.
    """
    x = np.asarray(x, dtype)
    W1, b1 = P["encoder_fc1.weight"].astype(dtype), P["encoder_fc1.bias"].astype(dtype)
    W2, b2 = P["encoder_fc2.weight"].astype(dtype), P["encoder_fc2.bias"].astype(dtype)
    Wih, Whh = P["lstm_cell.weight_ih"].astype(dtype), P["lstm_cell.weight_hh"].astype(dtype)
    W3, b3 = P["decoder_fc1.weight"].astype(dtype), P["decoder_fc1.bias"].astype(dtype)
    W4, b4 = P["decoder_fc2.weight"].astype(dtype), P["decoder_fc2.bias"].astype(dtype)
    N, T, _ = x.shape
    a1 = np.maximum(x @ W1.T + b1, 0)                    # agent.py:49 (inner relu)
    e = a1 @ W2.T + b2                                   # [N,T,128]  (same for fw and bw: :49-50)
    gx = e @ Wih.T                                       # input-side gate pre-activations, shared
    # direction d=0 consumes frames 0..T-1, d=1 consumes T-1..0 (agent.py:49-52)
    hs = np.zeros((2, N, T, H), dtype)                   # h after consuming frame t (indexed by frame)
    cs = np.zeros((2, N, T, H), dtype)
    gates = np.zeros((2, N, T, 4 * H), dtype)            # post-activation i,f,g,o
    for d in range(2):
        h = np.zeros((N, H), dtype)
        c = np.zeros((N, H), dtype)
        order = range(T) if d == 0 else range(T - 1, -1, -1)
        for t in order:
            pre = gx[:, t] + h @ Whh.T
            i, f = _sig(pre[:, :H]), _sig(pre[:, H:2 * H])
            g, o = np.tanh(pre[:, 2 * H:3 * H]), _sig(pre[:, 3 * H:])
            c = f * c + i * g
            h = o * np.tanh(c)
            hs[d, :, t], cs[d, :, t] = h, c
            gates[d, :, t] = np.concatenate([i, f, g, o], 1)
    hcat = np.maximum(np.concatenate([hs[0], hs[1]], 2), 0)   # relu on the concat (agent.py:62)
    d1 = np.maximum(hcat @ W3.T + b3, 0)
    q = (d1 @ W4.T + b4)[..., 0]
    if keep:
        return q, dict(x=x, a1=a1, e=e, hs=hs, cs=cs, gates=gates, hcat=hcat, d1=d1)
    return q


def brain_backward(P, cache, dq, dtype=np.float32):
    """Hand-derived gradient of sum(q*dq) w.r.t. the 10 parameter tensors (BPTT through the shared
    cell: both directions accumulate into the same weight_ih / weight_hh)."""
    W2 = P["encoder_fc2.weight"].astype(dtype)
    Wih, Whh = P["lstm_cell.weight_ih"].astype(dtype), P["lstm_cell.weight_hh"].astype(dtype)
    W3, W4 = P["decoder_fc1.weight"].astype(dtype), P["decoder_fc2.weight"].astype(dtype)
    x, a1, e = cache["x"], cache["a1"], cache["e"]
    hs, cs, gates, hcat, d1 = cache["hs"], cache["cs"], cache["gates"], cache["hcat"], cache["d1"]
    N, T, _ = x.shape
    dq = np.asarray(dq, dtype)
    G = {}
    G["decoder_fc2.weight"] = np.einsum("nt,nth->h", dq, d1)[None]
    G["decoder_fc2.bias"] = dq.sum().reshape(1)
    dd1 = dq[..., None] * W4[0] * (d1 > 0)
    G["decoder_fc1.weight"] = np.einsum("nth,ntk->hk", dd1, hcat)
    G["decoder_fc1.bias"] = dd1.sum((0, 1))
    dhcat = (dd1 @ W3) * (hcat > 0)
    dH = np.stack([dhcat[..., :H], dhcat[..., H:]], 0)       # [2,N,T,H]
    dG = np.zeros((2, N, T, 4 * H), dtype)                   # grad wrt gate pre-activations
    for d in range(2):
        dh_rec = np.zeros((N, H), dtype)
        dc = np.zeros((N, H), dtype)
        order = list(range(T)) if d == 0 else list(range(T - 1, -1, -1))
        for s in range(T - 1, -1, -1):
            t = order[s]
            i, f = gates[d, :, t, :H], gates[d, :, t, H:2 * H]
            g, o = gates[d, :, t, 2 * H:3 * H], gates[d, :, t, 3 * H:]
            c = cs[d, :, t]
            c_prev = cs[d, :, order[s - 1]] if s > 0 else np.zeros((N, H), dtype)
            tc = np.tanh(c)
            dh = dH[d, :, t] + dh_rec
            dc = dc + dh * o * (1 - tc * tc)
            dpre = np.concatenate([dc * g * i * (1 - i), dc * c_prev * f * (1 - f),
                                   dc * i * (1 - g * g), dh * tc * o * (1 - o)], 1)
            dG[d, :, t] = dpre
            dh_rec = dpre @ Whh
            dc = dc * f
    # h_prev for each (d, frame): the state before consuming that frame
    hprev = np.zeros_like(hs)
    hprev[0, :, 1:] = hs[0, :, :-1]
    hprev[1, :, :-1] = hs[1, :, 1:]
    G["lstm_cell.weight_hh"] = np.einsum("dntg,dnth->gh", dG, hprev)
    dgx = dG[0] + dG[1]
    G["lstm_cell.weight_ih"] = np.einsum("ntg,nth->gh", dgx, e)
    de = dgx @ Wih
    G["encoder_fc2.weight"] = np.einsum("nth,ntk->hk", de, a1)
    G["encoder_fc2.bias"] = de.sum((0, 1))
    da1 = (de @ W2) * (a1 > 0)
    G["encoder_fc1.weight"] = np.einsum("nth,ntk->hk", da1, x)
    G["encoder_fc1.bias"] = da1.sum((0, 1))
    return G


def build_states(batch, dtype=np.float32):
    """[B,1,T] float64 collated columns -> state/new_state [B,T,2] fp32 (models/agent.py:107-129)."""
    B = np.asarray(batch["action"]).shape[0]
    f = lambda k: np.asarray(batch[k]).reshape(B, -1).astype(np.float32).astype(dtype)
    state = np.stack([f("old_state_iou"), f("annotated_frames")], 2)
    new_state = np.stack([f("new_state_iou"), f("next_annotated_frames")], 2)
    return state, new_state


def dqn_targets(P_policy, P_target, new_state, reward_step, reward_done, gamma, dtype=np.float32):
    """Double-DQN targets (models/agent.py:133-141): a*=argmax policy(s'), Qn=target(s')[a*]."""
    q_pol = brain_forward(P_policy, new_state, dtype)
    a_star = q_pol.argmax(1)                              # first max, as torch.max(1)[1] on CPU
    q_tgt = brain_forward(P_target, new_state, dtype)
    qn = q_tgt[np.arange(len(a_star)), a_star]
    y_step = qn * dtype(gamma) + reward_step.astype(np.float32).astype(dtype) * dtype(0.1)
    y_done = reward_done.astype(np.float32).astype(dtype) * dtype(0.1)
    return y_step, y_done, a_star


def dqn_loss_and_grads(P_policy, P_target, batch, gamma, dtype=np.float32):
    """loss = mse(Qsa, y_step) + mse(Qsa, y_done) (models/agent.py:144-151) and dL/dtheta."""
    state, new_state = build_states(batch, dtype)
    action = np.asarray(batch["action"]).reshape(-1).astype(np.int64)
    B = action.shape[0]
    y_step, y_done, _ = dqn_targets(P_policy, P_target, new_state, np.asarray(batch["reward_step"]).reshape(-1),
                                    np.asarray(batch["reward_done"]).reshape(-1), gamma, dtype)
    q, cache = brain_forward(P_policy, state, dtype, keep=True)
    qsa = q[np.arange(B), action]
    loss = np.mean((qsa - y_step) ** 2, dtype=dtype) + np.mean((qsa - y_done) ** 2, dtype=dtype)
    dq = np.zeros_like(q)
    dq[np.arange(B), action] = (dtype(2.0) / dtype(B)) * ((qsa - y_step) + (qsa - y_done))
    return dtype(loss), brain_backward(P_policy, cache, dq, dtype)


def clamp_adam(P, G, M, V, step, lr, wd, beta1=0.9, beta2=0.999, eps=1e-8, clamp=1.0):
    """In-place: grad clamp to [-1,1] (models/agent.py:157-159) then torch.optim.Adam's update with
    coupled L2 (g += wd*p after the clamp), lerp-form first moment, ``denom = sqrt(v)/sqrt(bc2)+eps``."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = np.float32(lr / bc1)
    bc2_sqrt = np.float32(np.sqrt(bc2))
    for k in P:
        g = np.clip(G[k].astype(np.float32), -clamp, clamp)
        g = g + np.float32(wd) * P[k]
        M[k] += (g - M[k]) * np.float32(1.0 - beta1)
        V[k] *= np.float32(beta2)
        V[k] += np.float32(1.0 - beta2) * g * g
        denom = np.sqrt(V[k]) / bc2_sqrt + np.float32(eps)
        P[k] -= step_size * (M[k] / denom)


def dqn_step(P_policy, P_target, M, V, step, batch, cfg, coin):
    """One Agent.update_agent (models/agent.py:103-166). ``coin`` is the np.random.random() draw
    for the hard target sync (:163-165). Mutates P_policy/P_target/M/V; returns loss."""
    loss, G = dqn_loss_and_grads(P_policy, P_target, batch, cfg["gamma"])
    clamp_adam(P_policy, G, M, V, step, cfg["lr"], cfg["weight_decay"])
    if coin < cfg["update_rate"]:
        for k in P_policy:
            P_target[k] = P_policy[k].copy()
    return float(loss)


def epsilon(steps_done, eps_start, eps_end, eps_decay):
    """models/agent.py:173-174."""
    return eps_end + (eps_start - eps_end) * np.exp(-0.5 * steps_done / eps_decay)
