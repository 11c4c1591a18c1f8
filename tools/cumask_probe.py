"""Is an HBM-bound fused kernel bound per CU or chip-wide?  Runs AssessNet (bf16, B = 128) on a stream restricted to half of the
CUs (hipExtStreamCreateWithCUMask) and on an unrestricted stream, and prints the per-layer table of both.  Tuning aid, GPU only."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivos_w_amd import _lib as L, synth  # noqa: E402
from ivos_w_amd.models.assessment import AssessNet  # noqa: E402

dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net = AssessNet(precision="bf16")
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.assessnet_state_dict(0).items()})
net.to(dev).eval()
tf, tp = synth.assess_inputs(8, seed=3)
ttf = torch.from_numpy(tf).to(dev).repeat(B // 8, 1, 1, 1).contiguous()
ttp = torch.from_numpy(tp).to(dev).repeat(B // 8, 1, 1).contiguous()
lib = L.lib()


def masked_stream(mask_words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(mask_words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(stream, tag):
    with torch.cuda.stream(stream):
        for _ in range(3):
            net(ttf, ttp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            net(ttf, ttp)
        e1.record(stream)
        torch.cuda.synchronize()
        print(f"{tag}: {e0.elapsed_time(e1) / 10:.3f} ms per {B}-frame forward")
        lib.ivosw_profile_start()
        for _ in range(3):
            net(ttf, ttp)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.ivosw_profile_report(buf, len(buf))
        t, c = ctypes.c_double(0), ctypes.c_int(0)
        lib.ivosw_profile_stop(ctypes.byref(t), ctypes.byref(c))
        print(buf.value.decode())


run(torch.cuda.Stream(dev), "all 256 CUs")
# 256 CUs = 8 words of 32 bits; the mask is in hardware CU order: try every other XCD's worth by alternating words
run(masked_stream([0xffffffff, 0, 0xffffffff, 0, 0xffffffff, 0, 0xffffffff, 0]), "mask words 0,2,4,6")
run(masked_stream([0x55555555] * 8), "mask every other CU")
