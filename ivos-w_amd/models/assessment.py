"""Segmentation-quality-assessment network on the MI355X — drop-in for the reference's ``models.assessment``.

Same class surface and ``state_dict`` (326 tensors) as /root/reference/models/assessment.py (``Encoder`` :12-63,
``AssessNet`` :66-182); ``forward(tf, tp)`` runs entirely in libivosw_hip.so:

  mask bbox (wavefront min/max, no D2H)  ->  fused ROI bilinear resample + (f-mean)/std to NHWC4
  ->  stem 7x7 (RGB|mask concatenated) + BN + ReLU  ->  max-pool  ->  16 bottlenecks as implicit-GEMM MFMA
  convolutions with folded BN / residual / ReLU epilogues  ->  8x8 average pool + fc1.

The torch modules below are parameter containers only (keys, shapes, checkpoint I/O); torchvision is not
needed and nothing is downloaded.  ``precision='fp32'`` (default) is the parity mode (fp32 MFMA, scores within
1e-4 rtol of the reference CPU path); ``precision='bf16x3'`` keeps fp32 tensors and the 1e-4 bar but runs every contraction
- the stem included - as three bf16 MFMA passes on (hi, lo) splits (a b ~ ah bh + ah bl + al bh: error ~ 2^-17 per product, 16 / 3 of
the fp32 matrix rate); ``precision='bf16'`` is the throughput mode (bf16 operands, fp32 accumulate).
Inference only (eval-mode BatchNorm): AssessNet training is outside the hot path.
"""
import ctypes
import os
import math

import torch
import torch.nn as nn

from .. import _lib as L


class _Bottleneck(nn.Module):
    """Parameter container with torchvision's ResNet bottleneck attribute names."""

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))
        else:
            self.downsample = None


def _stage(inplanes, planes, blocks, stride):
    layers = [_Bottleneck(inplanes, planes, stride, True)]
    layers += [_Bottleneck(planes * 4, planes, 1, False) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class Encoder(nn.Module):
    """ResNet-50 trunk + the extra 1-channel stems (conv1_m / conv1_n are unused upstream but part of the
    checkpoint, models/assessment.py:15-20)."""

    def __init__(self):
        super().__init__()
        self.conv1_m = nn.Conv2d(1, 64, 7, 2, 3, bias=True)
        self.conv1_p = nn.Conv2d(1, 64, 7, 2, 3, bias=False)
        self.conv1_n = nn.Conv2d(1, 64, 7, 2, 3, bias=False)
        for m in (self.conv1_m, self.conv1_p, self.conv1_n):
            fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2.0 / fan))
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.res2 = _stage(64, 64, 3, 1)
        self.res3 = _stage(256, 128, 4, 2)
        self.res4 = _stage(512, 256, 6, 2)
        self.res5 = _stage(1024, 512, 3, 2)
        for m in [self.conv1] + [c for s in (self.res2, self.res3, self.res4, self.res5) for c in s.modules()
                                 if isinstance(c, nn.Conv2d)]:
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")   # no pretrained download
        self.register_buffer("mean", torch.FloatTensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.FloatTensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def forward(self, in_f, in_p, in_g=None):
        raise RuntimeError("Encoder has no standalone forward on the MI355X path; call AssessNet(tf, tp)")


_DTYPES = {"fp32": L.F32, "bf16": L.BF16, "bf16x3": L.F32X3}
_TAPS = {"roi": (1, (256, 256, 4)), "stem": (2, (128, 128, 64)), "pool": (3, (64, 64, 64)),
         "res2": (4, (64, 64, 256)), "res3": (5, (32, 32, 512)), "res4": (6, (16, 16, 1024)),
         "res5": (7, (8, 8, 2048)), "pooled": (8, (2048,))}


class AssessNet(nn.Module):
    def __init__(self, precision="fp32", chunk=0):
        super().__init__()
        if precision not in _DTYPES:
            raise ValueError("precision must be 'fp32', 'bf16x3' or 'bf16'")
        self.Encoder = Encoder()
        self.fc1 = nn.Linear(2048, 1)
        self.cnt = 0
        self.precision = precision
        self.chunk = chunk
        self._packed = None
        self._packed_key = None
        self._wver = 0                      # bumped whenever the weights can have changed (load_state_dict, .to(), ...)
        self._ws = L.Workspace()

    # ------------------------------------------------------------------ weights
    def invalidate_packed(self):
        """Call after modifying parameters in place by hand (load_state_dict / .to() / .float() do it themselves)."""
        self._wver += 1

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._wver += 1
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._wver += 1
        return out

    def _weights_key(self):
        # O(1): a forward at eval sizes (B ~ 100 frames) is ~1.6 ms of GPU time, walking the 326 tensors of the
        # state_dict on every call cost a comparable amount of host time.  The first and last tensors stand guard
        # against in-place edits that bypass the counter.
        w0, w1 = self.Encoder.conv1.weight, self.fc1.weight
        key = (self.precision, self._wver, str(w1.device), w0.data_ptr(), w0._version, w1.data_ptr(), w1._version)
        if os.environ.get("IVOSW_WEIGHTS_KEY", "") == "full":
            # debug / safety switch: walk every tensor, so an in-place edit of ANY parameter or BN buffer that bypassed
            # load_state_dict / .to() / invalidate_packed() re-packs the weights (the O(1) key watches two tensors and a counter)
            key += tuple((t.data_ptr(), t._version) for t in self.state_dict().values())
        return key

    def _ensure_packed(self):
        key = self._weights_key()
        if self._packed is not None and key == self._packed_key:
            return self._packed
        dev = self.fc1.weight.device
        lib, dt = L.lib(), _DTYPES[self.precision]
        sd = self.state_dict()
        assert len(sd) == L.ASSESS_NTENSORS
        keep, ptrs = [], (ctypes.c_void_p * L.ASSESS_NTENSORS)()
        for i, (k, t) in enumerate(sd.items()):
            if not t.is_floating_point():              # num_batches_tracked
                ptrs[i] = None
                continue
            t32 = t.detach().to(torch.float32).contiguous()
            keep.append(t32)
            ptrs[i] = L.dptr(t32).value
        packed = torch.empty(int(lib.ivosw_assess_packed_bytes(dt)), dtype=torch.uint8, device=dev)
        L.check(lib.ivosw_assess_pack(L.dptr(packed), dt, ptrs, L.ASSESS_NTENSORS, L.stream_ptr(dev)), "assess_pack")
        torch.cuda.current_stream(dev).synchronize()   # `keep` temporaries may be freed after this
        self._packed, self._packed_key = packed, key
        return packed

    # ------------------------------------------------------------------ forward
    def _run(self, tf, tp, tap=None):
        if self.training:
            raise RuntimeError("AssessNet on the MI355X path is inference-only (eval-mode BatchNorm); call .eval() — "
                               "training AssessNet is outside the hot path")
        tf = tf.detach().to(torch.float32).contiguous()
        tp = tp.detach().to(torch.float32).contiguous()
        B, C, H, W = tf.shape
        assert C == 3 and tuple(tp.shape) == (B, H, W)
        dev = tf.device
        packed = self._ensure_packed()
        lib, dt = L.lib(), _DTYPES[self.precision]
        chunk = B if tap else self.chunk
        nbytes = lib.ivosw_assess_ws_bytes(dt, B, H, W, chunk)
        ws = self._ws.get(nbytes, dev)
        scores = torch.empty(B, dtype=torch.float32, device=dev)
        stage, tap_t = 0, None
        if tap:
            stage, shp = _TAPS[tap]
            tdt = torch.float32 if (tap == "pooled" or self.precision != "bf16") else torch.bfloat16
            tap_t = torch.empty((B,) + shp, dtype=tdt, device=dev)
        L.check(lib.ivosw_assess_forward(L.dptr(packed), dt, L.dptr(tf), L.dptr(tp), B, H, W, L.dptr(scores), L.dptr(ws),
                                         nbytes, chunk, stage, L.dptr(tap_t) if tap else None, L.stream_ptr(dev)),
                "assess_forward")
        return scores, tap_t

    def forward(self, tf, tp):
        """tf [B,3,H,W] in [0,1], tp [B,H,W] soft mask -> quality [B,1] ((1,) when B == 1, like the
        reference's ``.squeeze()``, models/assessment.py:179)."""
        scores, _ = self._run(tf, tp)
        return scores if scores.shape[0] == 1 else scores[:, None]

    def forward_objects(self, all_F, all_P, n_objects):
        """Scores of every (object, frame) unit of ONE video without replicating the frames: all_F [n,3,H,W] fp32 on the
        device, all_P [n,C,H,W] fp32 on the device (any layout whose [H,W] planes are contiguous, e.g. the object-major
        ProbStore view), channel i+1 = object i (utils/utils_agent.py:118-119).  Returns [n_objects, n] fp32 on the device."""
        if self.training:
            raise RuntimeError("AssessNet on the MI355X path is inference-only; call .eval()")
        n, C3, H, W = all_F.shape
        assert C3 == 3 and all_P.shape[0] == n and tuple(all_P.shape[2:]) == (H, W) and all_P.shape[1] > n_objects
        if all_F.dtype != torch.float32 or not all_F.is_contiguous():
            all_F = all_F.detach().to(torch.float32).contiguous()
        if all_P.dtype != torch.float32 or all_P.stride(3) != 1 or all_P.stride(2) != W:
            all_P = all_P.detach().to(torch.float32).contiguous()
        dev = all_F.device
        packed = self._ensure_packed()
        lib, dt = L.lib(), _DTYPES[self.precision]
        units = n * n_objects
        nbytes = lib.ivosw_assess_ws_bytes(dt, units, H, W, self.chunk)
        ws = self._ws.get(nbytes, dev)
        scores = torch.empty(n_objects, n, dtype=torch.float32, device=dev)
        masks = all_P[:, 1:]                                # a view: channel 0 is the background
        L.check(lib.ivosw_assess_forward_objects(L.dptr(packed), dt, L.dptr(all_F), n, ctypes.c_void_p(masks.data_ptr()),
                                                 all_P.stride(0), all_P.stride(1), n_objects, H, W, L.dptr(scores),
                                                 L.dptr(ws), nbytes, self.chunk, L.stream_ptr(dev)), "assess_forward_objects")
        return scores

    def forward_tap(self, tf, tp, tap):
        """Debug/test hook: also returns one intermediate (NHWC; see ``_TAPS``)."""
        return self._run(tf, tp, tap)

    def all2yxhw(self, mask, scale=1.5):
        """Binary mask [B,H,W] -> (y,x,h,w) [B,4] on device (models/assessment.py:110-161)."""
        if scale != 1.5:
            raise ValueError("the HIP bbox kernel implements the scale=1.5 the reference forward uses")
        m = mask.detach().to(torch.float32).contiguous()
        B, H, W = m.shape
        out = torch.empty(B, 4, dtype=torch.float32, device=m.device)
        scratch = torch.empty(B, 4, dtype=torch.int32, device=m.device)
        L.check(L.lib().ivosw_mask_bbox(L.dptr(m), B, H, W, L.dptr(out), L.dptr(scratch), L.stream_ptr(m.device)),
                "mask_bbox")
        return out
