"""DAVIS J / F metrics on the device — drop-in for ``davisinteractive.metrics.batched_jaccard`` /
``batched_f_measure`` as the reference's ``sequence_metric`` uses them (utils/misc.py:118-162).

The label maps go to the GPU once (or already live there as the VOS model's argmax output); two kernels
(``csrc/metrics.hip``) return six integer counts per (frame, object), and the float64 ratios are formed here with the
package's own expressions — so J and F equal the CPU implementation bit for bit whenever the counts do.
There is no CPU fallback: without the HIP library or a GPU the call raises.
"""
import numpy as np
import torch

from . import _lib as L

_ws = {}


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("ivos_w_amd.metrics: no GPU — the J/F kernels have no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _as_labels(a, dev):
    """[N,H,W] integer labels (numpy or tensor, any integer dtype) -> contiguous uint8 device tensor."""
    if isinstance(a, torch.Tensor):
        t = a if a.is_cuda else a.to(dev, non_blocking=True)
        if t.dtype != torch.uint8:
            t = t.to(torch.uint8)
        return t.contiguous()
    a = np.asarray(a)
    if a.dtype != np.uint8:
        if a.size and (a.min() < 0 or a.max() > 255):
            raise ValueError("label maps must hold values in 0..255")
        a = a.astype(np.uint8)
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _check(y_true, y_pred):
    if y_true.ndim != 3:
        raise ValueError(f"y_true array must have 3 dimensions. Found {y_true.ndim} dimensions")
    if y_pred.ndim != 3:
        raise ValueError(f"y_pred array must have 3 dimensions. Found {y_pred.ndim} dimensions")
    if tuple(y_true.shape) != tuple(y_pred.shape):
        raise ValueError(f"y_true and y_pred must have the same shape. {tuple(y_true.shape)} != {tuple(y_pred.shape)}")


def _object_ids(gt, nb_objects):
    if nb_objects is None:
        ids = torch.unique(gt)
        ids = ids[(ids < 255) & (ids > 0)].cpu().numpy().astype(np.int64)
    else:
        ids = np.asarray([i + 1 for i in range(nb_objects)], dtype=np.int64)
    if len(ids) == 0:
        raise ValueError("Number of objects in y_true should be higher than 0.")
    return ids


def bound_pixels(shape, bound_th=0.008):
    return bound_th if bound_th >= 1 else np.ceil(bound_th * np.linalg.norm(shape))


def jf_counts(y_true, y_pred, nb_objects=None, bound_th=0.008):
    """-> (ids [O], counts int64 [N,O,6]): |gt&pred|, |gt|pred|, #pred-boundary, #gt-boundary, matched pred, matched gt."""
    _check(y_true, y_pred)
    dev = y_true.device if isinstance(y_true, torch.Tensor) and y_true.is_cuda else _device()
    gt, pr = _as_labels(y_true, dev), _as_labels(y_pred, dev)
    ids = _object_ids(gt, nb_objects)
    if ids.max() > 255:
        raise ValueError("object ids above 255 cannot occur in uint8 label maps")
    N, H, W = gt.shape
    bp = bound_pixels((H, W), bound_th)
    if bp != int(bp) or bp > 32:
        raise ValueError(f"boundary tolerance {bp} px: the kernel supports integer radii up to 32")
    lib = L.lib()
    counts = torch.empty((N, len(ids), 6), dtype=torch.int64, device=dev)
    nbytes = lib.ivosw_jf_ws_bytes(N, H, W, len(ids))
    key = (dev.index, "jf")
    ws = _ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    idb = bytes(ids.astype(np.uint8).tolist())
    L.check(lib.ivosw_jf_counts(L.dptr(gt, torch.uint8), L.dptr(pr, torch.uint8), N, H, W, idb, len(ids), int(bp),
                                L.dptr(counts, torch.int64), L.dptr(ws, torch.uint8), nbytes, L.stream_ptr(dev)), "jf_counts")
    return ids, counts.cpu().numpy()


def _jaccard_from(counts):
    inter, union = counts[..., 0], counts[..., 1]
    out = np.empty(inter.shape, dtype=np.float64)
    for idx in np.ndindex(inter.shape):
        out[idx] = 1.0 if np.isclose(union[idx], 0) else inter[idx] / union[idx]
    return out


def _f_from(counts):
    out = np.empty(counts.shape[:-1], dtype=np.float64)
    for idx in np.ndindex(out.shape):
        n_fg, n_gt, m_fg, m_gt = (counts[idx][k] for k in (2, 3, 4, 5))
        if n_fg == 0 and n_gt > 0:
            precision, recall = 1, 0
        elif n_fg > 0 and n_gt == 0:
            precision, recall = 0, 1
        elif n_fg == 0 and n_gt == 0:
            precision, recall = 1, 1
        else:
            precision = m_fg / float(n_fg)
            recall = m_gt / float(n_gt)
        out[idx] = 0 if precision + recall == 0 else 2 * precision * recall / (precision + recall)
    return out


def batched_jaccard(y_true, y_pred, average_over_objects=True, nb_objects=None):
    _, counts = jf_counts(y_true, y_pred, nb_objects)
    j = _jaccard_from(counts)
    return j.mean(axis=1) if average_over_objects else j


def batched_f_measure(y_true, y_pred, average_over_objects=True, nb_objects=None, bound_th=0.008):
    _, counts = jf_counts(y_true, y_pred, nb_objects, bound_th)
    f = _f_from(counts)
    return f.mean(axis=1) if average_over_objects else f


def batched_j_and_f(y_true, y_pred, average_over_objects=True, nb_objects=None, bound_th=0.008):
    """Both metrics from ONE pass over the label maps (the reference runs the two functions back to back)."""
    _, counts = jf_counts(y_true, y_pred, nb_objects, bound_th)
    j, f = _jaccard_from(counts), _f_from(counts)
    if average_over_objects:
        j, f = j.mean(axis=1), f.mean(axis=1)
    return j, f
