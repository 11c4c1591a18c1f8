"""ORACLE / CPU BASELINE (test infrastructure — never imported by the product path).

The "reference CPU path" of SURVEY §8(d) for the Double-DQN step: stock PyTorch-CPU operators
(``nn.Linear`` / ``nn.LSTMCell`` / autograd / ``optim.Adam``) assembled by the build's own module definitions, on all
host cores.  The reference itself cannot travel to the GPU box, so this is what ``bench.py``'s ``dqn.cpu_baseline``
times (kind "port"); it follows the reference's structure operator for operator so that it is in the same speed
class (BASELINE.md measured the imported reference at 6.8 steps/s on 8 threads):

  * ``TorchBrain``      <- Brain              /root/reference/models/agent.py:13-64  (python loop over T, encoder recomputed
                                               for both directions, in-place ReLUs)
  * ``TorchDQN.update`` <- Agent.update_agent models/agent.py:103-166 (3 forwards, two-MSE loss, backward, clamp, Adam,
                                               coin-flip target sync)

Pinned by tests/test_oracle_brain.py::test_torch_cpu_baseline_matches_reference_golden against the goldens recorded
from the imported reference (tests/golden/dqn_steps.npz, brain_forward.npz).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class TorchBrain(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder_fc1 = nn.Linear(2, 128)
        self.encoder_fc2 = nn.Linear(128, 128)
        self.lstm_cell = nn.LSTMCell(128, 128, False)
        self.decoder_fc1 = nn.Linear(256, 128)
        self.decoder_fc2 = nn.Linear(128, 1)

    def encode(self, x):
        return self.encoder_fc2(F.relu(self.encoder_fc1(x)))

    def forward(self, inp):                                     # inp [N,T,2]
        T = inp.shape[1]
        fw, bw = [], []
        sf = sb = None
        for t in range(T):                                      # agent.py:46-54: both directions inside one loop
            sf = self.lstm_cell(self.encode(inp[:, t]), sf)
            sb = self.lstm_cell(self.encode(inp[:, T - 1 - t]), sb)
            fw.append(sf[0])
            bw.append(sb[0])
        bw = bw[::-1]
        qs = []
        for t in range(T):                                      # agent.py:58-63
            h = F.relu(torch.cat([fw[t], bw[t]], 1))
            qs.append(self.decoder_fc2(F.relu(self.decoder_fc1(h))))
        return torch.cat(qs, 1)


class TorchDQN:
    def __init__(self, P_policy, P_target, gamma=0.95, lr=5e-6, weight_decay=5e-4, update_rate=0.05):
        self.policy, self.target = TorchBrain(), TorchBrain()
        self.policy.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P_policy.items()})
        self.target.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P_target.items()})
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=lr, weight_decay=weight_decay)
        self.gamma = torch.tensor(gamma, dtype=torch.float32)
        self.update_rate = update_rate

    def loss_and_grads(self, batch):
        """Forward + backward of Agent.update_agent (models/agent.py:103-160) up to, and without, the clamp: (loss tensor, nothing
        stepped); gradients are left in the policy parameters' .grad."""
        B = len(batch["action"])
        col = lambda k: torch.as_tensor(np.asarray(batch[k])).reshape(B, -1).float()
        state = torch.stack([col("old_state_iou"), col("annotated_frames")], 2)
        new_state = torch.stack([col("new_state_iou"), col("next_annotated_frames")], 2)
        action = torch.as_tensor(np.asarray(batch["action"])).reshape(B, 1).long()
        r_step = torch.as_tensor(np.asarray(batch["reward_step"])).reshape(B).float()
        r_done = torch.as_tensor(np.asarray(batch["reward_done"])).reshape(B).float()
        with torch.no_grad():
            a_star = self.policy(new_state).max(1)[1].view(B, 1)
            q_next = self.target(new_state).gather(1, a_star).view(B)
        y_step = self.gamma * q_next + 0.1 * r_step
        y_done = 0.1 * r_done
        q_sa = self.policy(state).gather(1, action).view(B)
        loss = F.mse_loss(q_sa, y_step) + F.mse_loss(q_sa, y_done)
        self.opt.zero_grad()
        loss.backward()
        return loss

    def grads(self):
        """{state_dict key: fp32 gradient} after loss_and_grads."""
        return {k: p.grad.detach().numpy().copy() for k, p in self.policy.named_parameters()}

    def update(self, batch, coin):
        """One Agent.update_agent on a collated minibatch (synth.collate_np layout); returns the loss."""
        loss = self.loss_and_grads(batch)
        for p in self.policy.parameters():
            p.grad.data.clamp_(-1, 1)
        self.opt.step()
        if coin < self.update_rate:
            self.target.load_state_dict(self.policy.state_dict())
        return float(loss.item())
