"""Deterministic synthetic weights and inputs (numpy ``RandomState`` legacy streams).

Nothing here is a checkpoint: there is no network, so every test, golden fixture and benchmark
regenerates weights and inputs from seeds.  ``RandomState`` is frozen by numpy policy, so the
container that produced ``tests/golden`` and the GPU box see identical bits.

Recipes follow SURVEY.md §8(c)/(d):
  * AssessNet: conv ~ N(0, sqrt(2/fan_out)); BN gamma~U(.5,1), beta~U(-.1,.1), running_mean~N(0,.1),
    running_var~U(.5,1.5); last BN gamma of every bottleneck x0.2 (bounded residual stream);
    fc1 ~ U(+-1/sqrt(2048)).  Key order == reference ``AssessNet.state_dict()`` (models/assessment.py:12-71).
  * Brain: every tensor ~ U(+-1/sqrt(fan_in)) (models/agent.py:13-31 default init, made reproducible).
  * A1/A2 inputs: uniform frames, soft-blob masks.  Q inputs: replay transitions, T=25.
"""
from collections import OrderedDict

import numpy as np

RESNET50_BLOCKS = (("res2", 3, 64, 1), ("res3", 4, 128, 2), ("res4", 6, 256, 2), ("res5", 3, 512, 2))

BRAIN_SHAPES = OrderedDict([
    ("encoder_fc1.weight", (128, 2)), ("encoder_fc1.bias", (128,)),
    ("encoder_fc2.weight", (128, 128)), ("encoder_fc2.bias", (128,)),
    ("lstm_cell.weight_ih", (512, 128)), ("lstm_cell.weight_hh", (512, 128)),
    ("decoder_fc1.weight", (128, 256)), ("decoder_fc1.bias", (128,)),
    ("decoder_fc2.weight", (1, 128)), ("decoder_fc2.bias", (1,)),
])
BRAIN_NPARAMS = sum(int(np.prod(s)) for s in BRAIN_SHAPES.values())  # 180 993


def brain_offsets():
    """name -> (offset, shape) into the flat fp32 parameter arena (state_dict order)."""
    out, off = OrderedDict(), 0
    for k, s in BRAIN_SHAPES.items():
        out[k] = (off, s)
        off += int(np.prod(s))
    return out


def brain_state_dict(seed=0):
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    fan_in = {"encoder_fc1": 2, "encoder_fc2": 128, "lstm_cell": 128, "decoder_fc1": 256, "decoder_fc2": 128}
    for k, s in BRAIN_SHAPES.items():
        b = 1.0 / np.sqrt(fan_in[k.split(".")[0]])
        sd[k] = rs.uniform(-b, b, size=s).astype(np.float32)
    return sd


def brain_flat(sd):
    return np.concatenate([np.asarray(sd[k], np.float32).ravel() for k in BRAIN_SHAPES])


def assessnet_key_shapes():
    """Ordered (key, shape, kind) for the 326 tensors of the reference AssessNet.state_dict()."""
    ks = [("Encoder.mean", (1, 3, 1, 1), "mean"), ("Encoder.std", (1, 3, 1, 1), "std"),
          ("Encoder.conv1_m.weight", (64, 1, 7, 7), "conv"), ("Encoder.conv1_m.bias", (64,), "cbias"),
          ("Encoder.conv1_p.weight", (64, 1, 7, 7), "conv"), ("Encoder.conv1_n.weight", (64, 1, 7, 7), "conv"),
          ("Encoder.conv1.weight", (64, 3, 7, 7), "conv")]

    def bn(prefix, c, last=False):
        return [(prefix + ".weight", (c,), "gamma_last" if last else "gamma"), (prefix + ".bias", (c,), "beta"),
                (prefix + ".running_mean", (c,), "rmean"), (prefix + ".running_var", (c,), "rvar"),
                (prefix + ".num_batches_tracked", (), "nbt")]

    ks += bn("Encoder.bn1", 64)
    inplanes = 64
    for name, nblk, planes, _stride in RESNET50_BLOCKS:
        for b in range(nblk):
            p = f"Encoder.{name}.{b}"
            ks.append((p + ".conv1.weight", (planes, inplanes, 1, 1), "conv"))
            ks += bn(p + ".bn1", planes)
            ks.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv"))
            ks += bn(p + ".bn2", planes)
            ks.append((p + ".conv3.weight", (planes * 4, planes, 1, 1), "conv"))
            ks += bn(p + ".bn3", planes * 4, last=True)
            if b == 0:
                ks.append((p + ".downsample.0.weight", (planes * 4, inplanes, 1, 1), "conv"))
                ks += bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    ks += [("fc1.weight", (1, 2048), "fc"), ("fc1.bias", (1,), "fc")]
    return ks


def assessnet_state_dict(seed=0, spread=False):
    """``spread``: a second recipe whose scores differ widely from frame to frame (the default recipe's last-BN gammas x 0.2 and
    default-size fc1 give nearly constant scores, which tests ranking decisions poorly): last-BN gammas x 0.5, fc1 weights x 6."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for k, s, kind in assessnet_key_shapes():
        if kind == "mean":
            v = np.array([0.485, 0.456, 0.406], np.float32).reshape(s)
        elif kind == "std":
            v = np.array([0.229, 0.224, 0.225], np.float32).reshape(s)
        elif kind == "conv":
            fan_out = s[0] * s[2] * s[3]
            v = (rs.standard_normal(s) * np.sqrt(2.0 / fan_out)).astype(np.float32)
        elif kind == "cbias":
            v = rs.uniform(-0.1, 0.1, s).astype(np.float32)
        elif kind in ("gamma", "gamma_last"):
            v = rs.uniform(0.5, 1.0, s).astype(np.float32)
            if kind == "gamma_last":
                v *= np.float32(0.5 if spread else 0.2)
        elif kind == "beta":
            v = rs.uniform(-0.1, 0.1, s).astype(np.float32)
        elif kind == "rmean":
            v = (rs.standard_normal(s) * 0.1).astype(np.float32)
        elif kind == "rvar":
            v = rs.uniform(0.5, 1.5, s).astype(np.float32)
        elif kind == "nbt":
            v = np.array(0, np.int64)
        elif kind == "fc":
            b = 1.0 / np.sqrt(2048.0)
            v = rs.uniform(-b, b, s).astype(np.float32)
            if spread and len(s) == 2:
                v *= np.float32(6.0)
        else:  # pragma: no cover
            raise KeyError(kind)
        sd[k] = v
    return sd


def assess_inputs(B, H=480, W=854, seed=1234, edge_cases=False, structured=False):
    """tf [B,3,H,W] fp32 in [0,1); tp [B,H,W] fp32 soft blob sigmoid((r0-dist)/8).

    ``structured`` adds smooth per-sample patterns (used by the parity fixtures; the benchmark keeps
    SURVEY's plain uniform frames).  ``edge_cases`` overrides the first samples with the masks the reference's bbox code treats
    specially (models/assessment.py:116-136): empty mask, tiny (<128 px) mask, border-touching mask,
    full-frame mask.
    """
    rs = np.random.RandomState(seed)
    tf = rs.rand(B, 3, H, W).astype(np.float32)
    if structured:
        # per-sample, per-channel plane waves under the noise so that pooled features differ between
        # samples (pure iid noise makes every frame look alike after global pooling) — parity tests only
        gy, gx = np.mgrid[0:H, 0:W].astype(np.float32)
        fx = rs.uniform(0.5, 6.0, (B, 3)).astype(np.float32)
        fy = rs.uniform(0.5, 6.0, (B, 3)).astype(np.float32)
        ph = rs.uniform(0.0, 6.28, (B, 3)).astype(np.float32)
        for b in range(B):
            for c in range(3):
                wave = np.sin(np.float32(6.2831853) * (fx[b, c] * gx / W + fy[b, c] * gy / H) + ph[b, c])
                tf[b, c] = np.clip(0.5 + 0.35 * wave + 0.3 * (tf[b, c] - 0.5), 0.0, 1.0)
    cy = rs.uniform(60, H - 60, B)
    cx = rs.uniform(100, W - 104, B)
    r0 = rs.uniform(20, 200, B)
    if B > 1:
        r0[1] = 30.0  # always one sample that triggers the min-128 rule
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    tp = np.empty((B, H, W), np.float32)
    for b in range(B):
        d = np.sqrt((yy - np.float32(cy[b])) ** 2 + (xx - np.float32(cx[b])) ** 2)
        with np.errstate(over="ignore"):
            tp[b] = 1.0 / (1.0 + np.exp(-(np.float32(r0[b]) - d) / 8.0))
    if edge_cases:
        n = 0
        if B > n:
            tp[n] = 0.0  # empty
            n += 1
        if B > n:
            tp[n] = 0.0
            tp[n, 200:210, 300:340] = 0.9  # tiny
            n += 1
        if B > n:
            tp[n] = 0.0
            tp[n, 0:100, 0:150] = 0.8  # touches top-left borders
            tp[n, H - 40:H, W - 60:W] = 0.7  # and bottom-right
            n += 1
        if B > n:
            tp[n] = 0.75  # everything foreground
            n += 1
    return tf, tp


def replay_transitions(n=50000, T=25, seed=2019):
    """Synthetic replay buffer (SURVEY §8d 'Q'): SoA dict of numpy arrays with the dtypes the
    reference DataLoader collation yields (datasets/agent_dataset.py:86-115)."""
    rs = np.random.RandomState(seed)
    old_iou = rs.uniform(0.2, 0.95, (n, T))
    new_iou = np.clip(old_iou + rs.uniform(0.0, 0.1, (n, T)), 0.0, 1.0)
    k = rs.randint(1, 5, n)
    annotated = np.zeros((n, T))
    for j in range(4):
        idx = rs.randint(0, T, n)
        sel = k > j
        np.add.at(annotated, (np.nonzero(sel)[0], idx[sel]), 1.0)
    action = rs.randint(0, T, n).astype(np.int64)
    nxt = annotated.copy()
    nxt[np.arange(n), action] += 1.0
    reward_step = np.where(rs.rand(n) < 0.8, 1, -1).astype(np.int64)
    reward_done = rs.standard_normal(n)
    done = (k == 4)
    return dict(action=action, reward_step=reward_step, reward_done=reward_done, done=done,
                old_state_iou=old_iou, new_state_iou=new_iou,
                annotated_frames=annotated, next_annotated_frames=nxt)


def minibatch_indices(step, n=50000, B=128, seed=7, rank=0):
    """Indices of minibatch ``step`` for ``rank`` (rank-offset stream)."""
    rs = np.random.RandomState(seed + 1000003 * rank)
    idx = None
    for _ in range(step + 1):
        idx = rs.randint(0, n, B)
    return idx


def brain_inputs(N, T, seed):
    """State tensors [N,T,2] float64: ch0 quality in [0,1), ch1 times-annotated in {0..3}."""
    rs = np.random.RandomState(seed)
    return np.stack([rs.rand(N, T), rs.randint(0, 4, (N, T)).astype(np.float64)], 2)


def collate_np(tr, idx):
    """Minibatch dict with the shapes/dtypes of the reference's collated DataLoader batch
    (datasets/agent_dataset.py:86-115 + default_collate): [B] int64/float64/bool, [B,1,T] float64."""
    out = {k: tr[k][idx] for k in ("action", "reward_step", "reward_done", "done")}
    for k in ("old_state_iou", "new_state_iou", "annotated_frames", "next_annotated_frames"):
        out[k] = tr[k][idx][:, None, :]
    return out


def label_maps(N, H=480, W=854, n_obj=3, seed=11, void=False):
    """Synthetic multi-object segmentation pair for the J/F metrics: `gt` = n_obj drifting ellipses (label o+1, later
    objects on top), `pred` = the same ellipses with per-frame jitter of centre / radii, so that IoU and boundary
    matches vary from frame to frame; frame 0 of `pred` is an exact copy, the last frame lacks the last object.
    Returns (gt, pred) uint8 [N,H,W]; `void` sprinkles a 255 (ignore) region into gt."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    gt = np.zeros((N, H, W), np.uint8)
    pred = np.zeros((N, H, W), np.uint8)
    cy, cx = rs.uniform(0.25, 0.75, n_obj) * H, rs.uniform(0.25, 0.75, n_obj) * W
    ry, rx = rs.uniform(0.05, 0.22, n_obj) * H, rs.uniform(0.05, 0.22, n_obj) * W
    for n in range(N):
        for o in range(n_obj):
            y0, x0 = cy[o] + 0.01 * H * n * (-1) ** o, cx[o] + 0.012 * W * n
            gt[n][((yy - y0) / ry[o]) ** 2 + ((xx - x0) / rx[o]) ** 2 <= 1.0] = o + 1
            if n == N - 1 and o == n_obj - 1 and N > 1:
                continue
            jy, jx = (0.0, 0.0) if n == 0 else rs.uniform(-0.02, 0.02, 2) * (H, W)
            sy, sx = (1.0, 1.0) if n == 0 else rs.uniform(0.9, 1.1, 2)
            pred[n][((yy - y0 - jy) / (ry[o] * sy)) ** 2 + ((xx - x0 - jx) / (rx[o] * sx)) ** 2 <= 1.0] = o + 1
    if void:
        gt[:, : max(1, H // 20), : max(1, W // 10)] = 255
    return gt, pred
