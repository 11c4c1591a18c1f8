"""Scenario drivers shared by the golden generator (tests/golden/make_goldens.py, which runs them against the IMPORTED
REFERENCE in the build container) and by the tests (which run them against the oracle / the product and compare with
the recorded fixtures).  Nothing here is reference source: a deterministic stand-in for the external MANet model, the
argument sets of the recorded ``get_results`` calls, and a list of checkpoint / meter / seed scenarios that takes the
module under test (the reference's ``utils.misc`` or ``ivos_w_amd.utils.misc``) as a parameter.
"""
import io
import os
import random
from collections import OrderedDict
from contextlib import redirect_stdout

import numpy as np
import torch


# ----------------------------------------------------------------------------- MANet stand-in (utils/utils_manet.py:59-163)
class FakeMANet:
    """Deterministic stand-in for the external MANet model: logits depend on the frame number, the embedding and the
    previous label, so a wrong propagation order or a wrong label hand-over changes the result.  Smooth logits: no
    near-ties between the two best classes (checked when the fixture is recorded)."""
    dynamic_seghead = None

    def __init__(self, C, hs, ws, dev):
        g = torch.Generator().manual_seed(3)
        self.basis = torch.randn(32, C, hs, ws, generator=g).to(dev)
        self.C, self.hs, self.ws = C, hs, ws
        self.calls = []

    def _logits(self, frame, emb, prev_label):
        x = self.basis[frame % 32] * 2.0 + emb.mean() * 0.1
        if prev_label is not None:
            pl = torch.nn.functional.interpolate(prev_label.float().reshape(1, 1, *prev_label.shape[-2:]), size=(self.hs, self.ws), mode="nearest")
            x = x + 0.5 * torch.nn.functional.one_hot(pl.long()[0, 0], self.C).permute(2, 0, 1).float()
        return x.unsqueeze(0)

    def int_seghead(self, ref_frame_embedding, ref_scribble_label, prev_round_label, global_map_tmp_dic, local_map_dics,
                    interaction_num, seq_names, gt_ids, frame_num, first_inter):
        self.calls.append(("int", frame_num[0]))
        return {seq_names[0]: self._logits(frame_num[0], ref_frame_embedding, None)}, local_map_dics

    def prop_seghead(self, ref_emb, prev_emb, cur_emb, scribble_label, prev_label, normalize_nearest_neighbor_distances,
                     use_local_map, seq_names, gt_ids, k_nearest_neighbors, global_map_tmp_dic, local_map_dics,
                     interaction_num, start_annotated_frame, frame_num, dynamic_seghead):
        self.calls.append(("prop", frame_num[0], k_nearest_neighbors))
        return {seq_names[0]: self._logits(frame_num[0], cur_emb, prev_label)}, global_map_tmp_dic, local_map_dics


SEG_KNNS = 5            # cfg.KNNS of the recorded runs (the reference reads it from its `config` module: utils_manet.py:101)

# name -> (n frames, channels, logits hs x ws, output h x w, next_frame)
SEG_CASES = OrderedDict([
    ("mid", (6, 3, 10, 18, 40, 72, 2)),          # interaction in the middle: both sweeps
    ("first", (5, 4, 12, 20, 36, 60, 0)),        # on the first frame: no backward sweep
    ("last", (4, 2, 9, 16, 27, 48, 3)),          # on the last frame: no forward sweep
    ("davis480p", (3, 4, 120, 214, 480, 854, 1)),    # MANet's stride-4 logits at 480p (recorded as slices + sums)
])


def seg_case(name, dev):
    """(model, kwargs of get_results without prev_label_storage) for a recorded case, tensors on `dev`."""
    n, C, hs, ws, h, w, nf = SEG_CASES[name]
    emb = torch.randn(n, 8, 6, 6, generator=torch.Generator().manual_seed(1)).to(dev)
    kw = dict(ref_frame_embedding=emb[nf:nf + 1], scribble_label=None, prev_label=None, eval_global_map_tmp_dic={}, local_map_dics=({}, {}),
              n_interaction=1, sequence="seq", obj_nums=C - 1, next_frame=nf, first_scribble=True, h=h, w=w, total_frame_num=n,
              embedding_memory=emb)
    return FakeMANet(C, hs, ws, dev), kw


SEG_SLICE = (slice(None), slice(None), slice(None, None, 37), slice(None, None, 41))      # what "davis480p" keeps of all_P


def seg_record(name, final_masks, all_P, storage, calls):
    """What a get_results run leaves behind, as arrays (small cases in full; the 480p case as strided slices, per-frame sums and
    the full label maps as uint8)."""
    fm = final_masks.detach().cpu().numpy()
    ap = all_P.detach().cpu().numpy()
    out = {f"{name}.final_masks_u8": fm.astype(np.uint8), f"{name}.final_is_float32": np.array(final_masks.dtype == torch.float32),
           f"{name}.calls": np.array([[0 if c[0] == "int" else 1, c[1], c[2] if len(c) > 2 else -1] for c in calls], np.int64),
           f"{name}.storage_keys": np.array(sorted(storage), np.int64),
           f"{name}.storage_is_int64": np.array(all(v.dtype == torch.int64 for v in storage.values())),
           f"{name}.storage_equals_final": np.array(all(np.array_equal(storage[k].detach().cpu().numpy().reshape(fm.shape[1:]), fm[k]) for k in storage))}
    if name == "davis480p":
        out[f"{name}.all_P_slices"] = ap[SEG_SLICE].copy()
        out[f"{name}.all_P_sums"] = ap.astype(np.float64).sum(axis=(2, 3))
    else:
        out[f"{name}.all_P"] = ap
    return out


# ----------------------------------------------------------------------------- checkpoint / meter / seed helpers (utils/misc.py:11-115)
class _RecNet:
    """Stands where a network stands in the checkpoint helpers: records what load_state_dict is handed."""

    def __init__(self, sd=None, fail=False):
        self._sd, self.fail, self.loaded = sd, fail, None

    def state_dict(self):
        return self._sd

    def load_state_dict(self, sd, strict=True):
        if self.fail:
            raise RuntimeError("size mismatch (injected)")
        self.loaded = {"keys": list(sd.keys()), "strict": bool(strict), "sums": [float(v.double().sum()) for v in sd.values()],
                       "ordered": isinstance(sd, OrderedDict)}


class _Holder:
    def __init__(self, net):
        self.policy_net = net


def _sd(keys):
    return OrderedDict((k, torch.full((2, 3), float(i + 1))) for i, k in enumerate(keys))


AGENT_KEY_SETS = OrderedDict([
    ("plain", ["encoder_fc1.weight", "encoder_fc1.bias", "lstm_fw.weight_ih", "decoder_fc2.bias"]),
    ("dataparallel", ["module.encoder_fc1.weight", "module.lstm_fw.weight_hh", "module.decoder_fc1.bias"]),
    ("video_match", ["base.conv1.weight", "module.base.layer1.0.bn1.running_mean", "encoder.base.conv1.weight", "fc.weight"]),
    ("module_inside", ["submodule.weight", "my_module_x.bias", "x.module"]),        # 'module' anywhere in the key triggers k[7:]
])
NETWORK_KEY_SETS = OrderedDict([
    ("plain", ["encoder.conv1.weight", "encoder.res2.0.bn1.running_var", "fc1.bias"]),
    ("dataparallel", ["module.encoder.conv1.weight", "module.fc1.weight"]),
    ("mixed", ["module.a.weight", "b.weight", "c.module.d"]),
])


def misc_scenarios(misc, tmp):
    """Runs the checkpoint / meter / seed scenarios against `misc` (the reference's utils.misc or the product's) and returns a
    JSON-able record: return values, the keys / strict flag / values handed to load_state_dict, the keys found in the files the
    save helpers write, what is printed."""
    rec = OrderedDict()

    def printed(fn):
        buf = io.StringIO()
        with redirect_stdout(buf):
            r = fn()
        return r, buf.getvalue().replace(str(tmp), "<tmp>")

    # ---- load_agent_checkpoint (:84-115)
    for name, keys in AGENT_KEY_SETS.items():
        for device in ("cpu", "cuda", "cuda:0", "torch.device:cuda", "torch.device:cpu"):
            d = os.path.join(tmp, f"agent_{name}")
            os.makedirs(d, exist_ok=True)
            torch.save(_sd(keys), os.path.join(d, "agent.pt"))
            dev = torch.device(device.split(":", 1)[1]) if device.startswith("torch.device") else device
            for strict in (False, True):
                net = _RecNet()
                r, out = printed(lambda: misc.load_agent_checkpoint(_Holder(net), d, device=dev, strict=strict))
                rec[f"load_agent/{name}/{device}/strict={strict}"] = {"ret": r, "loaded": net.loaded, "printed": out}
    d = os.path.join(tmp, "agent_plain")
    net = _RecNet()
    r, out = printed(lambda: misc.load_agent_checkpoint(_Holder(net), d))             # defaults: device='cpu', strict=False
    rec["load_agent/defaults"] = {"ret": r, "loaded": net.loaded, "printed": out}
    r, out = printed(lambda: misc.load_agent_checkpoint(None, d))
    rec["load_agent/agent_none"] = {"ret": r, "printed": out}
    r, out = printed(lambda: misc.load_agent_checkpoint(_Holder(_RecNet()), os.path.join(tmp, "nowhere")))
    rec["load_agent/missing_file"] = {"ret": r, "printed": out}
    r, out = printed(lambda: misc.load_agent_checkpoint(_Holder(_RecNet(fail=True)), d))
    rec["load_agent/load_raises"] = {"ret": r, "printed": out}
    with open(os.path.join(tmp, "agent_plain", "agent.pt"), "wb") as f:
        f.write(b"not a checkpoint")
    r, out = printed(lambda: misc.load_agent_checkpoint(_Holder(_RecNet()), d))
    rec["load_agent/corrupt_file"] = {"ret": r, "printed": out}

    # ---- load_network_checkpoint (:53-72)
    for name, keys in NETWORK_KEY_SETS.items():
        path = os.path.join(tmp, f"net_{name}.pt")
        torch.save(_sd(keys), path)
        for device in ("cpu", "gpu", "cuda"):
            for strict in (True, False):
                net = _RecNet()
                r, out = printed(lambda: misc.load_network_checkpoint(path, encoder=net, device=device, strict=strict))
                rec[f"load_network/{name}/{device}/strict={strict}"] = {"ret": r, "loaded": net.loaded, "printed": out}
    net = _RecNet()
    r, out = printed(lambda: misc.load_network_checkpoint(os.path.join(tmp, "net_plain.pt"), encoder=net))      # defaults
    rec["load_network/defaults"] = {"ret": r, "loaded": net.loaded, "printed": out}
    r, out = printed(lambda: misc.load_network_checkpoint(os.path.join(tmp, "nope.pt"), encoder=None))
    rec["load_network/missing_file"] = {"ret": r, "printed": out}
    try:
        misc.load_network_checkpoint(os.path.join(tmp, "net_plain.pt"), encoder=_RecNet(fail=True))
        rec["load_network/load_raises"] = "returned"
    except RuntimeError as e:                                   # this loader does NOT swallow (:53-72 has no try)
        rec["load_network/load_raises"] = "raised: " + str(e)

    # ---- save helpers (:42-49, :75-88): what is in the file
    def file_record(path):
        sd = torch.load(path, map_location="cpu")
        return {"exists": True, "keys": list(sd.keys()), "sums": [float(v.double().sum()) for v in sd.values()],
                "devices": sorted({str(v.device) for v in sd.values()})}
    for name, keys in (("plain", AGENT_KEY_SETS["plain"]), ("dataparallel", AGENT_KEY_SETS["dataparallel"])):
        d = os.path.join(tmp, f"save_{name}")
        r1 = misc.save_agent_checkpoint(_RecNet(_sd(keys)), d)
        r2 = misc.save_agent_checkpoint(_RecNet(_sd(keys)), d, epoch=7)
        r3 = misc.save_network_checkpoint(os.path.join(d, "deeper", "dir"), _RecNet(_sd(keys)))
        rec[f"save/{name}"] = {"ret": [r1, r2, r3], "files": sorted(os.listdir(d)), "agent.pt": file_record(os.path.join(d, "agent.pt")),
                               "agent_epoch_7.pt": file_record(os.path.join(d, "agent_epoch_7.pt")),
                               "assess_net.pt": file_record(os.path.join(d, "deeper", "dir", "assess_net.pt"))}

    # ---- AverageMeter (:18-38)
    m = misc.AverageMeter()
    trace = [[m.val, m.avg, m.sum, m.count]]
    for v, n in ((2.0, 2), (5.0, 1), (0.25, 4)):
        m.update(v, n) if n != 1 else m.update(v)
        trace.append([m.val, m.avg, m.sum, m.count])
    m.reset()
    trace.append([m.val, m.avg, m.sum, m.count])
    rec["average_meter"] = trace

    # ---- set_random_seed (:11-15): the three host generators it seeds
    misc.set_random_seed(3)
    a = [float(np.random.random()), float(random.random()), float(torch.rand(1).item())]
    misc.set_random_seed(3)
    b = [float(np.random.random()), float(random.random()), float(torch.rand(1).item())]
    rec["set_random_seed"] = {"first": a, "again": b}
    return rec
