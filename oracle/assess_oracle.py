"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement of the reference's segmentation-quality-assessment forward:

  * ``mask_bbox_yxhw``  <- AssessNet.all2yxhw     /root/reference/models/assessment.py:110-161  (numpy, fp64 like the reference)
  * ``roi_theta``       <- AssessNet.get_ROI_grid  models/assessment.py:75-93  (fp32; the unused inverse grid :95-107 is dropped)
  * ``roi_sample``      <- F.affine_grid + F.grid_sample(bilinear, zeros, align_corners=True)  models/assessment.py:104-105,173-174
  * ``encoder_forward`` <- Encoder.forward         models/assessment.py:46-63 + torchvision ResNet-50 v1.5 topology
                           (third-party, un-vendored: torchvision 0.4.x per README.md:30 — restated from its published
                           architecture: Bottleneck x[3,4,6,3], stride on the 3x3, 1x1+BN downsample; eval-mode BN eps=1e-5)
  * ``assess_forward``  <- AssessNet.forward       models/assessment.py:164-182

Floating-point convolutions use torch's own CPU operators (the same arithmetic the reference dispatches to);
the integer/coordinate logic is numpy.  Pinned by tests/golden/assess_*.npz recorded from the imported
reference (tests/golden/make_goldens.py).  The torchvision boundary itself is "parity unpinned" upstream
(the reference has no tests); our pin is the state_dict key/shape list + goldens through torch's Conv2d/BatchNorm2d.
"""
import numpy as np
import torch
import torch.nn.functional as F

BLOCKS = (("res2", 3, 1), ("res3", 4, 2), ("res4", 6, 2), ("res5", 3, 2))


def mask_bbox_yxhw(tm, scale=1.5):
    """tm [B,H,W] binary -> [B,4] float32 (y,x,h,w).  Integer min/max, min extent 128 with int(res/2)
    truncation, x1.5 growth in float64, clamp to [-5, dim+5]; empty mask => whole frame using H,W (not H-1,W-1)."""
    tm = np.asarray(tm)
    B, Hh, Ww = tm.shape
    out = np.zeros((B, 4), np.float32)
    for b in range(B):
        fg = tm[b] >= 0.49
        rows = np.flatnonzero(fg.any(1))
        cols = np.flatnonzero(fg.any(0))
        if rows.size == 0:
            y0, y1, x0, x1 = 0, Hh, 0, Ww
        else:
            y0, y1, x0, x1 = int(rows[0]), int(rows[-1]), int(cols[0]), int(cols[-1])
        if y1 - y0 < 128:
            half = int((128.0 - (y1 - y0)) / 2)
            y0, y1 = y0 - half, y1 + half
        if x1 - x0 < 128:
            half = int((128.0 - (x1 - x0)) / 2)
            x0, x1 = x0 - half, x1 + half
        oh, ow = y1 - y0 + 1, x1 - x0 + 1
        k = (scale - 1) / 2.0
        fy0, fy1 = max(-5.0, y0 - k * oh), min(Hh + 5.0, y1 + k * oh)
        fx0, fx1 = max(-5.0, x0 - k * ow), min(Ww + 5.0, x1 + k * ow)
        out[b] = [(fy1 + fy0) / 2.0, (fx1 + fx0) / 2.0, fy1 - fy0 + 1, fx1 - fx0 + 1]
    return out


def roi_theta(yxhw, Hh, Ww):
    """fp32 [B,4]: theta00, theta02, theta11, theta12 of the forward affine (models/assessment.py:79-92)."""
    r = np.asarray(yxhw, np.float32)
    two = np.float32(2.0)
    ymin, ymax = r[:, 0] - r[:, 2] / two, r[:, 0] + r[:, 2] / two
    xmin, xmax = r[:, 1] - r[:, 3] / two, r[:, 1] + r[:, 3] / two
    wm, hm = np.float32(Ww - 1), np.float32(Hh - 1)
    return np.stack([(xmax - xmin) / wm, (xmin + xmax - wm) / wm,
                     (ymax - ymin) / hm, (ymin + ymax - hm) / hm], 1).astype(np.float32)


def _linspace_m1_1(n):
    """torch.linspace(-1, 1, n) in fp32 as affine_grid builds it (symmetric halves)."""
    step = np.float32(2.0) / np.float32(n - 1)
    i = np.arange(n)
    lo = np.float32(-1.0) + i.astype(np.float32) * step
    hi = np.float32(1.0) - (n - 1 - i).astype(np.float32) * step
    return np.where(i < n // 2, lo, hi).astype(np.float32)


def roi_sample(img, theta4, out_hw=(256, 256)):
    """img [B,C,H,W] fp32 -> [B,C,256,256]: bilinear, zero padding, align_corners=True."""
    img = np.asarray(img, np.float32)
    B, C, Hh, Ww = img.shape
    oh, ow = out_hw
    u, v = _linspace_m1_1(ow), _linspace_m1_1(oh)
    out = np.zeros((B, C, oh, ow), np.float32)
    half = np.float32(0.5)
    for b in range(B):
        gx = u * theta4[b, 0] + theta4[b, 1]
        gy = v * theta4[b, 2] + theta4[b, 3]
        sx = ((gx + np.float32(1)) * half) * np.float32(Ww - 1)
        sy = ((gy + np.float32(1)) * half) * np.float32(Hh - 1)
        x0f, y0f = np.floor(sx), np.floor(sy)
        x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)
        wx1, wy1 = sx - x0f, sy - y0f
        wx0, wy0 = np.float32(1) - wx1, np.float32(1) - wy1
        acc = np.zeros((C, oh, ow), np.float32)
        for dy, wy in ((0, wy0), (1, wy1)):
            yy = y0 + dy
            vy = (yy >= 0) & (yy < Hh)
            yyc = np.clip(yy, 0, Hh - 1)
            for dx, wx in ((0, wx0), (1, wx1)):
                xx = x0 + dx
                vx = (xx >= 0) & (xx < Ww)
                xxc = np.clip(xx, 0, Ww - 1)
                w = (wy * vy)[:, None] * (wx * vx)[None, :]
                acc += img[b][:, yyc][:, :, xxc] * w.astype(np.float32)
        out[b] = acc
    return out


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def encoder_forward(sd, f_roi, p_roi, taps=None):
    """sd: dict key->torch fp32 tensor (reference keys).  Returns r5 [B,2048,8,8]."""
    f = (f_roi - sd["Encoder.mean"]) / sd["Encoder.std"]
    x = F.conv2d(f, sd["Encoder.conv1.weight"], None, 2, 3) + \
        F.conv2d(p_roi[:, None], sd["Encoder.conv1_p.weight"], None, 2, 3)
    x = F.relu(_bn(x, sd, "Encoder.bn1"))
    if taps is not None:
        taps["stem"] = x
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps["pool"] = x
    for name, nblk, stride in BLOCKS:
        for b in range(nblk):
            p = f"Encoder.{name}.{b}"
            s = stride if b == 0 else 1
            y = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
            y = F.relu(_bn(F.conv2d(y, sd[p + ".conv2.weight"], None, s, 1), sd, p + ".bn2"))
            y = _bn(F.conv2d(y, sd[p + ".conv3.weight"]), sd, p + ".bn3")
            if b == 0:
                x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, s), sd, p + ".downsample.1")
            x = F.relu(y + x)
        if taps is not None:
            taps[name] = x
    return x


def to_torch_sd(sd_np):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}


def assess_forward(sd, tf, tp, taps=None):
    """sd: torch state dict; tf [B,3,H,W], tp [B,H,W] numpy fp32 -> scores [B] fp32 (the reference returns
    [B,1], or (1,) for B=1 because of .squeeze(), models/assessment.py:179-182)."""
    tf, tp = np.asarray(tf, np.float32), np.asarray(tp, np.float32)
    tm = (tp > 0.5).astype(np.float32)
    yxhw = mask_bbox_yxhw(tm, 1.5)
    th = roi_theta(yxhw, tf.shape[2], tf.shape[3])
    f_roi = roi_sample(tf, th)
    p_roi = roi_sample(tp[:, None], th)[:, 0]
    if taps is not None:
        taps.update(yxhw=yxhw, theta=th, f_roi=f_roi, p_roi=p_roi)
    with torch.no_grad():
        r5 = encoder_forward(sd, torch.from_numpy(f_roi), torch.from_numpy(p_roi), taps)
        pooled = F.avg_pool2d(r5, 8).flatten(1)
        if taps is not None:
            taps["pooled"] = pooled
        out = F.linear(pooled, sd["fc1.weight"], sd["fc1.bias"])
    return out[:, 0].numpy()
