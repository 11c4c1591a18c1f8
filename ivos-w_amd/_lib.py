"""ctypes binding of libivosw_hip.so — the only way the host classes reach the arithmetic.

No CPU fallback exists: ``lib()`` raises if the shared library is missing, and ``dptr`` rejects
non-CUDA tensors, so a product call without the HIP extension fails loudly.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libivosw_hip.so")

F32, BF16, F32X3 = 0, 1, 2
BRAIN_NPARAMS = 180993
ASSESS_NTENSORS = 326

_p, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

SIGNATURES = {
    "ivosw_last_error": (C.c_char_p, []),
    "ivosw_version": (_i, []),
    "ivosw_brain_ws_bytes": (_sz, [_i, _i]),
    "ivosw_brain_forward": (_i, [_p, _p, _i, _i, _p, _p, _sz, _p]),
    "ivosw_brain_argmax": (_i, [_p, _i, _i, _p, _p]),
    "ivosw_dqn_ws_bytes": (_sz, [_i, _i]),
    "ivosw_dqn_loss_grad": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p, _sz, _p]),
    "ivosw_clamp_adam": (_i, [_p, _p, _p, _p, _i, _i, _f, _f, _f, _f, _f, _f, _f, _p]),
    "ivosw_adam_state_bytes": (_sz, []),
    "ivosw_clamp_adam_dev": (_i, [_p, _p, _p, _p, _i, _p, _f, _f, _f, _f, _f, _f, _f, _p]),
    "ivosw_copy_f32": (_i, [_p, _p, _sz, _p]),
    "ivosw_p2p_arena_bytes": (_sz, [_i, _sz]),
    "ivosw_p2p_handle_bytes": (_sz, []),
    "ivosw_p2p_alloc": (_i, [_sz, C.POINTER(_p), _p, _sz]),
    "ivosw_p2p_open": (_i, [_p, C.POINTER(_p)]),
    "ivosw_p2p_close": (_i, [_p]),
    "ivosw_p2p_free": (_i, [_p]),
    "ivosw_p2p_error": (_i, [_p, C.POINTER(_i)]),
    "ivosw_p2p_allreduce": (_i, [_p, _p, _i, _i, _i, C.POINTER(_p), C.c_uint, _i, _p]),
    "ivosw_p2p_allreduce_clamp_adam": (_i, [_p, _p, _i, _i, _i, C.POINTER(_p), C.c_uint, _i, _p, _p, _p, _i] + [_f] * 6 + [_p]),
    "ivosw_replay_gather": (_i, [_p] * 8 + [_i, _i] + [_p] * 5 + [_p]),
    "ivosw_replay_draw_state_bytes": (_sz, []),
    "ivosw_replay_draw_index": (C.c_ulonglong, [C.c_ulonglong, C.c_uint, C.c_uint, _i]),
    "ivosw_replay_draw_gather": (_i, [_p] * 8 + [_i, _i, _i] + [_p] * 6 + [_p]),
    "ivosw_dqn_step_drawn": (_i, [_p] * 10 + [_i, _i, _i, _f] + [_p] * 9 + [_sz] + [_p] * 3 + [_f] * 7 + [_p]),
    "ivosw_mask_bbox": (_i, [_p, _i, _i, _i, _p, _p, _p]),
    "ivosw_roi_sample": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "ivosw_assess_packed_bytes": (_sz, [_i]),
    "ivosw_assess_pack": (_i, [_p, _i, C.POINTER(_p), _i, _p]),
    "ivosw_assess_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ivosw_assess_split": (_i, [_i, _i, _i]),
    "ivosw_assess_forward": (_i, [_p, _i, _p, _p, _i, _i, _i, _p, _p, _sz, _i, _i, _p, _p]),
    "ivosw_assess_forward_objects": (_i, [_p, _i, _p, _i, _p, C.c_long, C.c_long, _i, _i, _i, _p, _p, _sz, _i, _p]),
    "ivosw_quality_state": (_i, [_p, _i, _i, _p, _p, _p, _p]),
    "ivosw_assess_dominant_kernel": (C.c_char_p, [_i]),
    "ivosw_jf_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "ivosw_jf_counts": (_i, [_p, _p, _i, _i, _i, C.c_char_p, _i, _i, _p, _p, _sz, _p]),
    "ivosw_seg_epilogue": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, C.c_long, C.c_long, _p, _p, _p, _p]),
    "ivosw_graph_begin": (_i, [_p]),
    "ivosw_graph_end": (_i, [_p, C.POINTER(_p), C.POINTER(_i)]),
    "ivosw_graph_launch": (_i, [_p, _p]),
    "ivosw_graph_destroy": (_i, [_p]),
    "ivosw_profile_start": (_i, []),
    "ivosw_profile_span_start": (_i, []),
    "ivosw_profile_span_stop": (_i, [C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(_i)]),
    "ivosw_profile_stop": (_i, [C.POINTER(C.c_double), C.POINTER(_i)]),
    "ivosw_profile_report": (_i, [C.c_char_p, _sz]),
    "ivosw_tune_set": (_i, [C.c_char_p, _i]),
    "ivosw_ablation_build": (_i, []),
    "ivosw_clock_probe": (_i, [_p, _i, _p]),
    "ivosw_assess_forget": (_i, [_p]),
}

# the tuning probes (include/ivosw_probe.h) live in a superset build of the same sources, libivosw_probe.so: never part of the product ABI
PROBE_LIB_PATH = os.path.join(_HERE, "libivosw_probe.so")
PROBE_SIGNATURES = {
    "ivosw_lstm_probe": (_i, [_p, _p]),
    "ivosw_bneck_probe": (_i, [_p] * 11 + [_i] * 5 + [_p, _p]),
    "ivosw_bneck_wide_probe": (_i, [_p] * 9 + [_i] * 5 + [_p, _p]),
    "ivosw_res2_stage_probe": (_i, [_p] * 4 + [_i] * 2 + [_p, _p]),
    "ivosw_gemm_bt_probe": (_i, [_p] * 4 + [_i] * 4 + [_p, _p]),
}

_lib = None
_probe = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"ivos_w_amd: {LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950). There is no CPU fallback for the hot path.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("IVOSW_BRINGUP_PARTIAL") and not hasattr(h, name):
                continue                    # bring-up only: a partially built library
            fn = getattr(h, name)           # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def probe_lib():
    """libivosw_probe.so: every entry of the product library plus the probes of include/ivosw_probe.h (tools/, one GPU test)."""
    global _probe
    if _probe is None:
        if not os.path.exists(PROBE_LIB_PATH):
            raise RuntimeError(f"ivos_w_amd: {PROBE_LIB_PATH} not found - build it with `python ivos-w_amd/build.py`")
        h = C.CDLL(PROBE_LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(PROBE_SIGNATURES.items()):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _probe = h
    return _probe


def use_probe_lib():
    """Tuning tools: route EVERY library call of this process through the probe build (the probes steer state inside the library - stamp
    pointers, tunables - that the product entries called afterwards read).  Call before anything else touches the library."""
    global _lib
    if _lib is not None and _lib is not _probe:
        raise RuntimeError("use_probe_lib() must run before the first library call of the process")
    _lib = probe_lib()
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"ivosw {what} failed ({rc}): {lib().ivosw_last_error().decode()}")


def tune_set(key, value):
    """ivosw_tune_set with its status checked: a switch that was silently not set would make an A/B test compare the default
    with itself."""
    if isinstance(key, str):
        key = key.encode()
    check(lib().ivosw_tune_set(key, int(value)), f"tune_set({key!r})")


def dptr(t, dtype=None):
    """Raw device pointer of a contiguous CUDA tensor (raises for CPU tensors: no fallback)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("ivos_w_amd: the hot path runs on the MI355X only — got a non-CUDA tensor "
                           "(move inputs/modules to the GPU; there is no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Workspace:
    """Grow-only byte workspace on one device (torch caching allocator owns the memory)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf


class Graph:
    """A captured launch sequence (ivosw_graph_*): ``with Graph.capture(device) as g: <ivosw calls via stream_ptr(device)>``
    records every library call made inside the block on a private stream into one HIP graph; ``g.launch()`` replays it on
    the caller's current stream.  No torch allocation may happen inside the block: hand in preallocated tensors."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.handle = None
        self.kernel_nodes = 0
        self._stream = None
        self._ctx = None

    @classmethod
    def capture(cls, device):
        return cls(device)

    def __enter__(self):
        self._stream = torch.cuda.Stream(self.device)
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        self._ctx = torch.cuda.stream(self._stream)
        self._ctx.__enter__()
        check(lib().ivosw_graph_begin(C.c_void_p(self._stream.cuda_stream)), "graph_begin")
        return self

    def __exit__(self, et, ev, tb):
        h, n = C.c_void_p(), C.c_int(0)
        rc = lib().ivosw_graph_end(C.c_void_p(self._stream.cuda_stream), C.byref(h), C.byref(n))
        self._ctx.__exit__(et, ev, tb)
        if et is None:
            check(rc, "graph_end")
            self.handle, self.kernel_nodes = h, n.value
        return False

    def launch(self):
        check(lib().ivosw_graph_launch(self.handle, stream_ptr(self.device)), "graph_launch")

    def __del__(self):
        if getattr(self, "handle", None):
            try:
                lib().ivosw_graph_destroy(self.handle)
            except Exception:
                pass
            self.handle = None
