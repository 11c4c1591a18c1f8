"""Seed / meter / checkpoint helpers with the reference's names and behaviour (utils/misc.py:11-115).

``sequence_metric`` (:118-162) keeps the reference's signature; its J / F measures run on the device
(``ivos_w_amd.metrics`` -> ``ivosw_jf_counts``) instead of the third-party ``davisinteractive`` package.
"""
import os
import random
from collections import OrderedDict

import numpy as np
import torch


def set_random_seed(seed):
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class AverageMeter:
    """Running average (val, sum, count, avg)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def _strip_module(k):
    return k[7:] if "module" in k else k


def save_network_checkpoint(ckpt_dir, assess_net):
    """Writes ``assess_net.pt`` = the state_dict as-is (the reference builds a stripped CPU copy but saves the
    original dict, utils/misc.py:42-49; the file content is therefore the plain state_dict)."""
    os.makedirs(ckpt_dir, exist_ok=True)
    torch.save(assess_net.state_dict(), os.path.join(ckpt_dir, "assess_net.pt"))


def load_network_checkpoint(ckpt_path, encoder=None, device="cpu", strict=True):
    """Returns False when the file is missing; strips 'module.' for device='cpu', adds it for device='gpu'."""
    if not os.path.exists(ckpt_path):
        return False
    sd = torch.load(ckpt_path, map_location="cpu")
    fixed = OrderedDict()
    for k, v in sd.items():
        if device == "cpu" and "module" in k:
            k = k[7:]
        elif device == "gpu" and "module" not in k:
            k = "module." + k
        fixed[k] = v
    encoder.load_state_dict(fixed, strict=strict)
    return True


def save_agent_checkpoint(net, ckpt_dir, epoch=None):
    os.makedirs(ckpt_dir, exist_ok=True)
    name = "agent.pt" if epoch is None else f"agent_epoch_{epoch}.pt"
    torch.save(OrderedDict((k, v.detach().clone().to("cpu")) for k, v in net.state_dict().items()),
               os.path.join(ckpt_dir, name))


def load_agent_checkpoint(agent, ckpt_dir, device="cpu", strict=False):
    """Never raises (utils/misc.py:99-115): returns None when agent/file is missing, 1 on success, -1 on failure."""
    if agent is None:
        return
    path = os.path.join(ckpt_dir, "agent.pt")
    if not os.path.exists(path):
        print(f"no model found in {path}")
        return
    print(f"load agent model from {path}")
    try:
        sd = torch.load(path, map_location="cpu")
        fixed = OrderedDict()
        for k, v in sd.items():
            k = _strip_module(k)
            if "base" in k and "encoder" not in k:
                k = "encoder." + k
            if str(device) == "cuda":
                k = "module." + k
            fixed[k] = v
        agent.policy_net.load_state_dict(fixed, strict=strict)
        return 1
    except Exception:
        print(f"catch some EXCEPTION when trying to load {path}")
        return -1


def sequence_metric(metric_to_optimize, gt_masks, pred_masks, nb_objects, average_over_objects=True,
                    convert_to_single_obj=False):
    """utils/misc.py:118-162 with the davisinteractive metrics computed on the device (ivos_w_amd.metrics).

    gt_masks / pred_masks: [N,H,W] integer label maps, numpy arrays (as the reference's callers pass them) or tensors
    (device tensors avoid the upload).  Returns the numpy float64 array the reference returns.  Like the reference,
    ``convert_to_single_obj`` rewrites the caller's arrays in place."""
    from .. import metrics

    if convert_to_single_obj:
        gt_masks[gt_masks > 0] = 1
        pred_masks[pred_masks > 0] = 1
        nb_objects = 1

    if metric_to_optimize == 'J':
        metric = metrics.batched_jaccard(gt_masks, pred_masks, average_over_objects=average_over_objects,
                                         nb_objects=nb_objects)
    elif metric_to_optimize == 'F':
        metric = metrics.batched_f_measure(gt_masks, pred_masks, average_over_objects=average_over_objects,
                                           nb_objects=nb_objects)
    elif metric_to_optimize == 'J_AND_F':
        jaccard, contour = metrics.batched_j_and_f(gt_masks, pred_masks, average_over_objects=average_over_objects,
                                                   nb_objects=nb_objects)
        metric = .5 * jaccard + .5 * contour
    return metric
