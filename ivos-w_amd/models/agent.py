"""Double-DQN frame-recommendation agent on the MI355X — drop-in for the reference's ``models.agent``.

Same class surface as /root/reference/models/agent.py (``Brain`` :13-64, ``Agent`` :67-236), with the
arithmetic replaced by libivosw_hip.so:

  Brain.forward        -> ivosw_brain_forward   (batched encoder/gate GEMMs + register-resident LSTM recurrence)
  Agent.update_agent   -> ivosw_dqn_loss_grad   (3 forwards, two-MSE Double-DQN loss, hand-derived BPTT)
                          [+ RCCL all-reduce of the flat gradient arena when torch.distributed is up]
                          ivosw_clamp_adam       (clamp [-1,1] + coupled-L2 Adam, one fused kernel)
                          ivosw_copy_f32         (hard target sync)
  Agent.action         -> ivosw_brain_forward + ivosw_brain_argmax (first max, like numpy)

torch modules (nn.Linear / nn.LSTMCell) are used only as parameter containers so that ``state_dict()`` keys,
shapes and default initialisation are the reference's; their ``forward`` is never called.  All ten tensors are
views into one flat fp32 arena (``Brain.flat``), which is what the C ABI, Adam and the all-reduce operate on.
"""
import math
import random

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .momory_pool import ReplayMemory

_ORDER = ("encoder_fc1.weight", "encoder_fc1.bias", "encoder_fc2.weight", "encoder_fc2.bias",
          "lstm_cell.weight_ih", "lstm_cell.weight_hh", "decoder_fc1.weight", "decoder_fc1.bias",
          "decoder_fc2.weight", "decoder_fc2.bias")


class Brain(nn.Module):
    def __init__(self, lstm_input_channels=128, hidden_channels=128, num_fc_concat=128):
        super().__init__()
        if (lstm_input_channels, hidden_channels, num_fc_concat) != (128, 128, 128):
            raise ValueError("the HIP Brain is specialised for the reference's 128/128/128 widths")
        self.input_channels, self.hidden_channels, self.num_fc_concat = 128, 128, 128
        # parameter containers, created in the reference's order so a seeded init matches it
        self.encoder_fc1 = nn.Linear(2, 128)
        self.encoder_fc2 = nn.Linear(128, 128)
        self.lstm_cell = nn.LSTMCell(128, 128, False)
        self.decoder_fc1 = nn.Linear(256, 128)
        self.decoder_fc2 = nn.Linear(128, 1)
        self.flat = None
        self.flat_grad = None
        self._ws = L.Workspace()
        self._pack()

    # ------------------------------------------------------------------ flat arena
    def _named(self):
        table = dict(self.named_parameters())
        return [table[k] for k in _ORDER]

    def _pack(self):
        """(Re)build the flat arena and make every parameter (and its .grad) a view into it."""
        params = self._named()
        dev = params[0].device
        flat = torch.empty(L.BRAIN_NPARAMS, dtype=torch.float32, device=dev)
        grad = torch.zeros(L.BRAIN_NPARAMS, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1).float())
            p.data = flat[off:off + n].view(p.shape)
            p.grad = grad[off:off + n].view(p.shape)
            off += n
        assert off == L.BRAIN_NPARAMS
        self.flat, self.flat_grad = flat, grad

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._pack()
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, input):
        """input [N,T,2] -> Q [N,T].  Inference only (the DQN update has its own fused backward)."""
        x = input.detach().to(dtype=torch.float32).contiguous()
        N, T, P = x.shape
        assert P == 2
        q = torch.empty(N, T, dtype=torch.float32, device=x.device)
        lib = L.lib()
        nbytes = lib.ivosw_brain_ws_bytes(N, T)
        ws = self._ws.get(nbytes, x.device)
        L.check(lib.ivosw_brain_forward(L.dptr(self.flat), L.dptr(x), N, T, L.dptr(q), L.dptr(ws), nbytes,
                                        L.stream_ptr(x.device)), "brain_forward")
        return q


class FusedClampAdam:
    """``optim.Adam(params, lr, weight_decay)`` + the reference's grad clamp, as one kernel over the flat arena."""

    def __init__(self, brain, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8, clamp=1.0):
        self.brain = brain
        self.param_groups = [dict(params=list(brain.parameters()), lr=lr, betas=betas, eps=eps,
                                  weight_decay=weight_decay, clamp=clamp)]
        self.state = dict(step=0, exp_avg=None, exp_avg_sq=None)
        self.grad_scale = 1.0

    def _ensure(self):
        flat = self.brain.flat
        if self.state["exp_avg"] is None or self.state["exp_avg"].device != flat.device:
            self.state["exp_avg"] = torch.zeros_like(flat)
            self.state["exp_avg_sq"] = torch.zeros_like(flat)

    def zero_grad(self, set_to_none=False):
        self.brain.flat_grad.zero_()

    def step(self):
        self._ensure()
        g = self.param_groups[0]
        self.state["step"] += 1
        b = self.brain
        L.check(L.lib().ivosw_clamp_adam(L.dptr(b.flat), L.dptr(b.flat_grad), L.dptr(self.state["exp_avg"]),
                                         L.dptr(self.state["exp_avg_sq"]), L.BRAIN_NPARAMS, self.state["step"],
                                         g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                                         g["clamp"], self.grad_scale, L.stream_ptr(b.flat.device)), "clamp_adam")

    # -- device-side step state (what a captured HIP graph replays; ivosw_clamp_adam_dev) -------------------------------
    def dev_state(self):
        """The 32-byte Adam step state on the device, (re)synchronised with the host step counter."""
        self._ensure()
        dev = self.brain.flat.device
        ds = self.state.get("dev")
        if ds is None or ds.device != dev:
            ds = torch.zeros(L.lib().ivosw_adam_state_bytes(), dtype=torch.uint8, device=dev)
            self.state["dev"], self.state["dev_step"] = ds, 0
        if self.state["dev_step"] != self.state["step"]:
            ds[16:20].copy_(torch.from_numpy(np.array([self.state["step"]], dtype=np.int32).view(np.uint8)))
            self.state["dev_step"] = self.state["step"]
        return ds

    def enqueue_dev_step(self):
        """ivosw_clamp_adam_dev on the current stream (inside a capture: recorded); the caller bumps the host counter per replay."""
        g, b = self.param_groups[0], self.brain
        L.check(L.lib().ivosw_clamp_adam_dev(L.dptr(b.flat), L.dptr(b.flat_grad), L.dptr(self.state["exp_avg"]),
                                             L.dptr(self.state["exp_avg_sq"]), L.BRAIN_NPARAMS, L.dptr(self.state["dev"]),
                                             g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], g["clamp"],
                                             self.grad_scale, L.stream_ptr(b.flat.device)), "clamp_adam_dev")

    def note_dev_steps(self, n=1):
        self.state["step"] += n
        self.state["dev_step"] = self.state["step"]

    def state_dict(self):
        self._ensure()
        return dict(state={k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.state.items() if not k.startswith("dev")},
                    param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self._ensure()
        self.state["step"] = int(sd["state"]["step"])
        self.state["exp_avg"].copy_(sd["state"]["exp_avg"])
        self.state["exp_avg_sq"].copy_(sd["state"]["exp_avg_sq"])


class Agent(nn.Module):
    def __init__(self, device, cfg):
        super().__init__()
        self.cfg = cfg
        self.device = device
        a = cfg.agent
        self.memory_size = a.memory_size
        self.GAMMA = a.gamma
        self.EPS_START, self.EPS_END, self.EPS_DECAY = a.eps_start, a.eps_end, a.eps_decay
        self.steps_done = 0
        self.update_rate = a.update_rate
        self.subset = cfg.data.subset
        self.memory_pool = ReplayMemory(self.memory_size)

        self.policy_net = Brain()
        self.target_net = Brain()
        self.target_net.load_state_dict(self.policy_net.state_dict())
        self.policy_net.to(self.device)
        self.target_net.to(self.device)

        self.loss = []
        self.loss_position = 0
        self.loss_capacity = 32
        self.loss_avg = 0
        self.optimizer = FusedClampAdam(self.policy_net, lr=a.lr, weight_decay=a.weight_decay)
        self._ws = L.Workspace()
        self._loss_dev = None

    # ------------------------------------------------------------------ data-parallel hook
    @staticmethod
    def _world():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist, dist.get_world_size()
        return None, 1

    # ------------------------------------------------------------------ DQN update
    def _device_batch(self, sample):
        """Collated DataLoader dict (datasets/agent_dataset.py) or a DeviceReplay.sample() dict -> device tensors."""
        dev = self.device
        if "state" in sample:                       # already gathered on device (ivosw_replay_gather)
            return (sample["state"], sample["new_state"], sample["action"], sample["reward_step"],
                    sample["reward_done"])
        B = sample["action"].shape[0]
        col = lambda k: torch.as_tensor(sample[k]).reshape(B, -1).to(torch.float32)
        state = torch.stack([col("old_state_iou"), col("annotated_frames")], 2).contiguous().to(dev)
        new_state = torch.stack([col("new_state_iou"), col("next_annotated_frames")], 2).contiguous().to(dev)
        action = torch.as_tensor(sample["action"]).reshape(B).to(torch.int64).to(dev)
        r_step = torch.as_tensor(sample["reward_step"]).reshape(B).to(torch.float32).to(dev)
        r_done = torch.as_tensor(sample["reward_done"]).reshape(B).to(torch.float32).to(dev)
        return state, new_state, action, r_step, r_done       # 'done' is loaded but unused upstream (agent.py:114)

    def loss_and_grads(self, sample):
        """Forward x3 + loss + backward into policy_net.flat_grad (unclamped). Returns the device loss scalar."""
        state, new_state, action, r_step, r_done = self._device_batch(sample)
        B, T, _ = state.shape
        lib = L.lib()
        nbytes = lib.ivosw_dqn_ws_bytes(B, T)
        ws = self._ws.get(nbytes, state.device)
        if self._loss_dev is None or self._loss_dev.device != state.device:
            self._loss_dev = torch.zeros(1, dtype=torch.float32, device=state.device)
        pn, tn = self.policy_net, self.target_net
        L.check(lib.ivosw_dqn_loss_grad(L.dptr(pn.flat), L.dptr(tn.flat), L.dptr(state), L.dptr(new_state),
                                        L.dptr(action, torch.int64), L.dptr(r_step), L.dptr(r_done), B, T,
                                        float(np.float32(self.GAMMA)), L.dptr(pn.flat_grad), L.dptr(self._loss_dev),
                                        L.dptr(ws), nbytes, L.stream_ptr(state.device)), "dqn_loss_grad")
        return self._loss_dev

    def update_agent(self, sample):
        if sample is None:
            print("no input")
            return
        loss = self.loss_and_grads(sample)
        self._update_avg_loss(loss)                 # synchronises (loss.item(), models/agent.py:166): the P2P error word rides with it
        self.apply_gradients()
        # hard target sync with probability update_rate; np.random is seeded identically on every rank
        if np.random.random() < self.update_rate:
            print("target_net updated!")
            self.sync_target()
        return self.loss[(self.loss_position - 1) % self.loss_capacity]

    def apply_gradients(self, check_every=1):
        """clamp + Adam on policy_net.flat_grad; with torch.distributed initialised: synchronous data parallel — the gradients are
        summed over the ranks first (RCCL over xGMI, or the opt-in one-shot P2P all-reduce fused with the update) and averaged
        inside the kernel, so the clamp sees the averaged gradient as a single large batch would."""
        dist, world = self._world()
        if world > 1 or dist is not None:           # an initialised process group of one rank (IVOSW_FORCE_DIST=1) takes the same path
            from .. import parallel
            parallel.data_parallel_step(self.policy_net, self.optimizer, check_every)
        else:
            self.optimizer.step()

    def sync_target(self):
        pn, tn = self.policy_net, self.target_net
        L.check(L.lib().ivosw_copy_f32(L.dptr(tn.flat), L.dptr(pn.flat), L.BRAIN_NPARAMS,
                                       L.stream_ptr(pn.flat.device)), "copy_f32")

    # ------------------------------------------------------------------ acting
    def action(self, state, verbose=True, device_out=None):
        """Reference surface: action(state [T,2] numpy, verbose) -> frame index (agent.py:168-196).  Extensions used by
        utils_agent's device-resident chain: `state` may be a [T,2] fp32 CUDA tensor, and with `device_out` (int64 [1] on
        the device) the greedy index is left THERE and None is returned, so the caller can fetch it together with its
        other results in one D2H copy (the epsilon branch still returns a host integer)."""
        self.steps_done += 1
        if self.cfg.phase != "train":
            eps_threshold = 0
        else:
            eps_threshold = self.EPS_END + (self.EPS_START - self.EPS_END) * \
                math.exp(-0.5 * self.steps_done / self.EPS_DECAY)
        on_device = torch.is_tensor(state) and state.is_cuda       # [T,2] fp32 already on the GPU (utils_agent's device chain)
        n_frames = state.shape[0] if on_device else np.asarray(state).shape[0]
        rand_flag = random.random()
        greedy = rand_flag > eps_threshold
        if verbose:
            print(f"step:{self.steps_done}, rand_flag:{rand_flag:.4f}, eps_threshold:{eps_threshold:.4f}, "
                  f"frame index was selected {'by agent' if greedy else 'randomly'}")
        if not greedy:
            return random.choice(np.array(range(n_frames)))
        st = state if on_device else torch.as_tensor(np.asarray(state), dtype=torch.float32).to(self.device)
        if device_out is not None:
            self.greedy_index_device(st, out=device_out)
            return None
        return np.int64(self.greedy_index_device(st).item())

    def greedy_index_device(self, state, out=None):
        """argmax_t Q(state)[t] for a device state [T,2] fp32, result left on the device (int64 [1], or written to `out`)."""
        q = self.policy_net(state[None])
        idx = out if out is not None else torch.empty(1, dtype=torch.int64, device=q.device)
        L.check(L.lib().ivosw_brain_argmax(L.dptr(q), 1, q.shape[1], L.dptr(idx), L.stream_ptr(q.device)), "argmax")
        return idx

    # ------------------------------------------------------------------ bookkeeping
    def _update_avg_loss(self, loss):
        self.note_loss(float(loss.detach().to("cpu").reshape(-1)[0]))

    def note_loss(self, value):
        """The 32-entry loss ring of the reference (models/agent.py:198-203), fed with a host float."""
        if len(self.loss) < self.loss_capacity:
            self.loss.append(None)
        self.loss[self.loss_position] = float(value)
        self.loss_position = (self.loss_position + 1) % self.loss_capacity
        self.loss_avg = sum(self.loss) / len(self.loss)

    def get_avg_loss(self):
        return self.loss_avg

    def set_train(self):
        self.policy_net.train()
        self.target_net.train()

    def set_eval(self):
        self.policy_net.eval()
        self.target_net.eval()

    def memory(self, state, old_frame, next_state, reward_step, reward_done, is_done, state_iou, next_state_iou,
               annotated_frames_str, next_annotated_frames_str, report_save_dir):
        self.memory_pool.push(state, old_frame, next_state, reward_step, reward_done, is_done, state_iou,
                              next_state_iou, annotated_frames_str, next_annotated_frames_str)
        self.memory_pool.push_to_csv(report_save_dir)


class CapturedDqnStep:
    """One Double-DQN training step as ONE HIP-graph launch (models/agent.py:128-160 minus the host coin flip):

        replay gather (minibatch indices read from ``self.idx`` on the device) -> 3 forwards + loss + BPTT
        -> [fused=True: clamp + Adam with the step counter on the device]

    With fused=False the graph stops at the gradients, for the data-parallel step (all-reduce, then the eager clamp+Adam).
    Replaces ~40 host launches (170-250 us of enqueue per step) by one hipGraphLaunch.  The arithmetic is the eager path's:
    the same entry points are recorded, so results are bit-identical to ``Agent.loss_and_grads`` + ``optimizer.step``."""

    def __init__(self, agent, replay, B, fused=True, draw_seed=None, steps=1, draw_state=None, capture=True):
        """draw_seed: None = the caller writes the minibatch rows into ``self.idx`` before every launch; an integer = the rows
        are drawn INSIDE the graph (``ivosw_replay_draw_gather``, uniform with replacement from a device-side counter-based
        generator seeded with it), ``self.idx`` then holds the rows of the last launch.  draw_state: share another captured
        step's generator state instead of creating one.  steps > 1 (needs the in-graph draw and fused=True): that many
        consecutive training steps per launch — everything a step changes (parameters, Adam moments and step counter, draw
        counter) lives on the device, so the recorded sequence simply repeats; the caller owns the host-side coin of the
        target sync, i.e. launches a multi-step graph only over steps whose coins do not fire (``GraphedDqnLoop``)."""
        dev = torch.device(agent.device)
        self.agent, self.replay, self.B, self.fused, self.steps = agent, replay, B, fused, int(steps)
        self.draw = draw_state if draw_state is not None else (replay.draw_state(draw_seed) if draw_seed is not None else None)
        if self.steps > 1 and (self.draw is None or not fused):
            raise ValueError("a multi-step graph needs the in-graph minibatch draw and the fused clamp + Adam")
        T = replay.T
        lib = L.lib()
        self.idx = torch.zeros(B, dtype=torch.int64, device=dev)
        self.state = torch.empty(B, T, 2, dtype=torch.float32, device=dev)
        self.new_state = torch.empty_like(self.state)
        self.action = torch.empty(B, dtype=torch.int64, device=dev)
        self.r_step = torch.empty(B, dtype=torch.float32, device=dev)
        self.r_done = torch.empty_like(self.r_step)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        nbytes = lib.ivosw_dqn_ws_bytes(B, T)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        pn, tn, opt = agent.policy_net, agent.target_net, agent.optimizer
        if fused:
            opt.dev_state()
        self._keys = (pn.flat.data_ptr(), tn.flat.data_ptr(), pn.flat_grad.data_ptr())
        self._hyper = self._hyper_now()          # lr / betas / eps / weight decay / clamp / grad_scale / gamma are baked into the graph
        self._nbytes, self.graph, self.kernel_nodes, self._onecall_args = nbytes, None, None, None
        if not capture:                          # capture=False: the same launches, enqueued plainly by launch() (LeanDqnLoop)
            return
        torch.cuda.synchronize(dev)
        with L.Graph.capture(dev) as g:
            self._enqueue()
        self.graph = g
        self.kernel_nodes = g.kernel_nodes

    def _enqueue(self):
        """The step's launches on the current stream (recorded when a capture is open)."""
        agent, replay, B, fused = self.agent, self.replay, self.B, self.fused
        dev = torch.device(agent.device)
        lib, T, nbytes = L.lib(), replay.T, self._nbytes
        pn, tn, opt = agent.policy_net, agent.target_net, agent.optimizer
        for _ in range(self.steps):
            st = L.stream_ptr(dev)
            r = replay
            if self.draw is not None and fused:
                # draw + gather folded into the encoder launch, the slab reduction into clamp + Adam: 8 kernel nodes per step instead of 10
                if self._onecall_args is None:   # every pointer and scalar is fixed for the life of the object (launch() checks): built once
                    g_, os_ = opt.param_groups[0], opt.state
                    self._onecall_args = (
                        L.dptr(pn.flat), L.dptr(tn.flat), L.dptr(r.old_iou), L.dptr(r.new_iou), L.dptr(r.ann), L.dptr(r.next_ann),
                        L.dptr(r.action), L.dptr(r.reward_step), L.dptr(r.reward_done), L.dptr(self.draw, torch.uint8), r.n, B, T,
                        float(np.float32(agent.GAMMA)), L.dptr(self.idx), L.dptr(self.state), L.dptr(self.new_state), L.dptr(self.action),
                        L.dptr(self.r_step), L.dptr(self.r_done), L.dptr(pn.flat_grad), L.dptr(self.loss), L.dptr(self.ws), nbytes,
                        L.dptr(os_["exp_avg"]), L.dptr(os_["exp_avg_sq"]), L.dptr(os_["dev"]), g_["lr"], g_["betas"][0], g_["betas"][1],
                        g_["eps"], g_["weight_decay"], g_["clamp"], opt.grad_scale)
                L.check(lib.ivosw_dqn_step_drawn(*self._onecall_args, st), "dqn_step_drawn")
                continue
            if self.draw is not None:
                r.sample_drawn(B, self.draw, out=dict(idx=self.idx, state=self.state, new_state=self.new_state, action=self.action,
                                                      reward_step=self.r_step, reward_done=self.r_done))
            else:
                L.check(lib.ivosw_replay_gather(L.dptr(r.old_iou), L.dptr(r.new_iou), L.dptr(r.ann), L.dptr(r.next_ann),
                                                L.dptr(r.action), L.dptr(r.reward_step), L.dptr(r.reward_done), L.dptr(self.idx), B, T,
                                                L.dptr(self.state), L.dptr(self.new_state), L.dptr(self.action), L.dptr(self.r_step),
                                                L.dptr(self.r_done), st), "replay_gather")
            L.check(lib.ivosw_dqn_loss_grad(L.dptr(pn.flat), L.dptr(tn.flat), L.dptr(self.state), L.dptr(self.new_state),
                                            L.dptr(self.action), L.dptr(self.r_step), L.dptr(self.r_done), B, T,
                                            float(np.float32(agent.GAMMA)), L.dptr(pn.flat_grad), L.dptr(self.loss),
                                            L.dptr(self.ws), nbytes, st), "dqn_loss_grad")
            if fused:
                opt.enqueue_dev_step()

    def _hyper_now(self):
        """What the captured launches bake in: gamma always (the loss); the optimizer's values only when clamp + Adam are part of
        the graph (fused) — the gradient-only graph of the data-parallel step leaves them to the eager update."""
        if not self.fused:
            return (float(self.agent.GAMMA),)
        g = self.agent.optimizer.param_groups[0]
        return (float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"]), float(g["clamp"]),
                float(self.agent.optimizer.grad_scale), float(self.agent.GAMMA))

    def launch(self):
        """Enqueue one step on the current stream; ``self.loss`` holds the device loss afterwards."""
        a = self.agent
        if self._keys != (a.policy_net.flat.data_ptr(), a.target_net.flat.data_ptr(), a.policy_net.flat_grad.data_ptr()):
            raise RuntimeError("the parameter arenas moved (.to() / re-pack) after capture: build a new CapturedDqnStep")
        if self._hyper != self._hyper_now():
            raise RuntimeError("a hyper-parameter (lr / betas / eps / weight_decay / clamp / grad_scale / gamma) changed after capture: "
                               "the graph replays the captured values - build a new CapturedDqnStep")
        if self.fused:
            a.optimizer.dev_state()              # resync if an eager step ran in between
        if self.graph is not None:
            self.graph.launch()
        else:
            self._enqueue()
        if self.fused:
            a.optimizer.note_dev_steps(self.steps)
        return self.loss


class GraphedDqnLoop:
    """The training loop of train_agent.py:177-182 / Agent.update_agent (models/agent.py:103-166) on captured graphs:
    every step = draw a minibatch, update the policy net, flip the reference's target-sync coin (``np.random.random() <
    update_rate``, one coin per step, in step order).  Steps are launched ``block`` at a time as ONE hipGraphLaunch whenever
    none of the block's coins fires (an 8.7 us bubble separates consecutive graph launches: per step it is 1/block of that);
    a block with a firing coin runs step by step with the sync where the reference has it.  Same coin stream, same minibatch
    stream (device counter), same arithmetic: results equal the step-by-step loop's bit for bit."""

    def __init__(self, agent, replay, B, draw_seed, block=8, draw_state=None):
        self.agent, self.block = agent, int(block)
        self.one = CapturedDqnStep(agent, replay, B, fused=True, draw_seed=draw_seed, draw_state=draw_state)
        self.many = CapturedDqnStep(agent, replay, B, fused=True, draw_state=self.one.draw, steps=self.block) if self.block > 1 else None
        self.launches = 0
        self.syncs = 0

    def run(self, n):
        """n training steps; returns the device loss tensor of the last one."""
        a = self.agent
        done, loss = 0, None
        while done < n:
            k = min(self.block, n - done)
            coins = np.random.random(k) < a.update_rate            # the same draws, in the same order, as k scalar calls
            if k == self.block and self.many is not None and not coins.any():
                loss = self.many.launch()
                self.launches += 1
            else:
                for c in coins:
                    loss = self.one.launch()
                    self.launches += 1
                    if c:
                        a.sync_target()
                        self.syncs += 1
            done += k
        return loss


class LeanDqnLoop:
    """The same loop from PLAIN launches out of preallocated buffers (ivosw_dqn_step_drawn: encoder with the draw + gather, recurrence,
    decoder, head, BPTT, two tail launches, clamp + Adam with the slab reduction — eight launches per step, no graph; ten before round 4).  With the launch chain this short the host keeps ahead of the GPU (~190 us of GPU
    work per step) and the step loses the bubble that separates two graph launches; on a loaded host the captured loop is the
    safer choice (``AutoDqnLoop`` measures).  Same arithmetic, same coin and minibatch streams: bit-identical to the other loops."""

    def __init__(self, agent, replay, B, draw_seed, draw_state=None):
        self.agent, self.replay, self.B = agent, replay, B
        self.step = CapturedDqnStep(agent, replay, B, fused=True, draw_seed=draw_seed, draw_state=draw_state, capture=False)
        self.draw = self.step.draw
        self.syncs = 0

    def run(self, n):
        a, loss = self.agent, None
        for _ in range(n):
            loss = self.step.launch()               # ivosw_dqn_step_drawn: eight plain launches, the step counters on the device
            if np.random.random() < a.update_rate:
                a.sync_target()
                self.syncs += 1
        return loss


class AutoDqnLoop:
    """Runs the first steps once through each launch mode (captured blocks / plain launches), timed, and the rest through the
    faster one.  The probe steps are ordinary training steps and the modes are bit-identical, so the trajectory does not depend
    on the choice."""

    def __init__(self, agent, replay, B, draw_seed, block=8, probe=256):
        self.agent, self.probe = agent, int(probe)
        self.lean = LeanDqnLoop(agent, replay, B, draw_seed)
        self.graphed = GraphedDqnLoop(agent, replay, B, draw_seed, block=block, draw_state=self.lean.draw)
        self.choice, self.probe_us = None, {}

    def run(self, n):
        import time
        dev = torch.device(self.agent.device)
        loss = None
        if self.choice is None and n >= 2 * self.probe:
            for name, loop in (("graph", self.graphed), ("plain", self.lean), ("graph", self.graphed), ("plain", self.lean)):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                loss = loop.run(self.probe // 2)
                torch.cuda.synchronize(dev)
                self.probe_us[name] = min(self.probe_us.get(name, 1e30), (time.perf_counter() - t0) / (self.probe // 2) * 1e6)
            self.choice = min(self.probe_us, key=self.probe_us.get)
            n -= 2 * self.probe
        loop = self.lean if self.choice == "plain" else self.graphed
        if n > 0:
            loss = loop.run(n)
        return loss

    @property
    def syncs(self):
        return self.lean.syncs + self.graphed.syncs
